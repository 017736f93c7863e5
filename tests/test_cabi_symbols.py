"""CPU-only: the C-ABI library builds, loads and exports every symbol include/*.h declares;
the host logic that needs no device behaves; and without a device the search path FAILS LOUDLY
instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sassy():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    import sassy_amd
    return sassy_amd


def declared_symbols():
    syms = set()
    for h in ("sassy.h", "sassy_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src):
            name = m.group(1)
            if name.startswith("sassy_") or name == "search":
                syms.add(name)
    return syms


def test_every_declared_symbol_is_exported(sassy):
    L = sassy.lib()
    decl = declared_symbols()
    assert {"sassy_searcher", "sassy_searcher_free", "search", "sassy_matches_free"} <= decl
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(L, name), name
    assert set(sassy.EXPORTED_SYMBOLS) >= decl


def test_match_struct_layout(sassy):
    # reference c/sassy.h:11-21: repr(C), 40 bytes, align 8
    assert C.sizeof(sassy.CMatch) == 40
    assert sassy.CMatch.cost.offset == 32 and sassy.CMatch.strand.offset == 36


def test_searcher_constructor_errors(sassy):
    with pytest.raises(sassy.SassyHipError, match="Unsupported alphabet"):
        sassy.Searcher("protein")
    # overhang: Iupac only, 0 <= alpha <= 1 (reference: Searcher::_overhang_check, src/search.rs:373-383)
    with pytest.raises(sassy.SassyHipError, match="[Oo]verhang"):
        sassy.Searcher("dna", rc=False, alpha=0.5)
    with pytest.raises(sassy.SassyHipError, match="Alpha"):
        sassy.Searcher("iupac", rc=False, alpha=1.5)
    sassy.Searcher("iupac", rc=False, alpha=0.5)
    # Ascii has no complement: like the reference, the searcher constructs and its first search fails
    # (src/c.rs:64 builds Searcher::<Ascii>::new(rc, ..); the panic comes from Profile::complement)
    with pytest.raises(sassy.SassyHipError, match="reverse complement is not defined"):
        sassy.Searcher("ascii", rc=True).search(b"abc", b"xxabcxx", 0)
    for a in ("dna", "DNA", "Iupac", "ascii"):
        sassy.Searcher(a, rc=False)


def test_required_halo(sassy):
    assert sassy.required_halo(32, 3) % 128 == 0
    assert sassy.required_halo(32, 3) >= 64 * 2
    assert sassy.required_halo(200, 20) >= 64 * 4 + 64


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_text_batch_marshalling(sassy):
    """sassy_amd.TextBatch (one buffer + offsets for search_many): the addresses and lengths the C call would get point at
    the texts -- from a list, from a buffer with a header in front; a text behind the buffer's end is refused (host logic,
    no device)."""
    import numpy as np
    texts = [b"ACGT", b"", b"TTTTTGGG", b"N" * 100]
    tb = sassy.TextBatch.from_list(texts)
    assert len(tb) == 4 and [C.string_at(int(a), int(l)) for a, l in zip(tb._addr, tb.lens)] == texts
    lens = np.array([len(t) for t in texts])
    starts = 5 + np.concatenate([[0], np.cumsum(lens)[:-1]])
    tb2 = sassy.TextBatch(b"#####" + b"".join(texts), starts, lens)
    assert [C.string_at(int(a), int(l)) for a, l in zip(tb2._addr, tb2.lens)] == texts
    assert len(sassy.TextBatch.from_list([])) == 0
    with pytest.raises(sassy.SassyHipError):
        sassy.TextBatch(b"ACGT", [2], [5])
    with pytest.raises(sassy.SassyHipError):
        sassy.TextBatch(b"ACGT", [0, 1], [1])


def test_seed_layout_of_the_seeded_search(sassy):
    """sassy_hip_seed_layout (host arithmetic of search_encoded_patterns' seeded search, no device): k + 1 disjoint seeds of
    at most two lengths and at most 10 rows inside the pattern, for every shape the path takes; plain patterns keep the
    even cut; CRISPR guides (20 bases + NGG) get their N into a 6-row seed instead of the even cut's 5-row one (expected
    table hits per position and guide 4.6e-3 -> 2.4e-3); a row that matches nothing ('X') may sit in a seed."""
    import random
    rng = random.Random(5)
    for m in range(8, 33):
        for k in range(0, 8):
            if m // (k + 1) < 3:
                continue
            for alphabet in ("dna", "iupac"):
                pats = [bytes(rng.choice(b"ACGT") for _ in range(m)) for _ in range(5)]
                if alphabet == "iupac":
                    pats = [bytes(rng.choice(b"ACGTNRYX") if rng.random() < 0.15 else c for c in p) for p in pats]
                lay = sassy.seed_layout(alphabet, pats, k)
                assert len(lay) == k + 1 and len({ln for _, ln in lay}) <= 2, (m, k, lay)
                last = 0
                for a, ln in sorted(lay):
                    assert a >= last and 1 <= ln <= 10 and a + ln <= m, (m, k, lay)
                    last = a + ln
    plain = [bytes(rng.choice(b"ACGT") for _ in range(23)) for _ in range(50)]
    assert sassy.seed_layout("iupac", plain, 3) == sassy.seed_layout("dna", plain, 3) == [(0, 6), (6, 6), (12, 6), (18, 5)]
    guides = [p[:20] + b"NGG" for p in plain]
    lay = sorted(sassy.seed_layout("iupac", guides, 3))

    def rate(layout):
        return sum(4.0 ** -(ln - (1 if a <= 20 < a + ln else 0)) for a, ln in layout)
    assert rate(lay) < 0.6 * rate([(0, 6), (6, 6), (12, 6), (18, 5)]) and abs(rate(lay) - 2.44e-3) < 1e-4, (lay, rate(lay))
    with pytest.raises(sassy.SassyHipError):
        sassy.seed_layout("dna", [b"ACGT"], 5)


def test_sub_piece_test_rows_of_the_seeded_search(sassy):
    """sassy_hip_seed_test_rows (host arithmetic, no device): for every shape and both seed layouts -- the even cut and a
    layout with gaps -- the k + 1 sub-pieces of a tested piece are disjoint, lie in the pattern outside the seed, start
    within 48 characters of the window (within 32: the narrow layout), keep their compared bits inside the 64 the kernel
    takes ((off & 15) + 2k + len <= 32), and sit on the seed's diagonal: off = win_left - (seed end - row) - k."""
    for m in range(8, 33):
        for k in range(0, 8):
            if m // (k + 1) < 3:
                continue
            even = sassy.seed_layout("dna", [b"A" * m], k)
            q = m // (k + 1)
            gaps = [(i * q + (1 if q > 3 else 0), min(q, 10) - (1 if q > 3 else 0)) for i in range(k + 1)]  # (a row left out in front of every seed)
            for seeds in (even, gaps):
                win, mx, rows = sassy.seed_test_rows(m, k, seeds)
                assert mx <= 47 and win <= 10 + 24, (m, k, seeds)
                tested = 0
                for p, subs in rows.items():
                    a0, l0 = seeds[p]
                    assert len(subs) == k + 1
                    tested += 1
                    last = 0
                    for a, ln, off in sorted(subs):
                        assert a >= last and ln >= 1 and a + ln <= m and (a + ln <= a0 or a >= a0 + l0), (m, k, seeds, p, subs)
                        assert (off & 15) + 2 * k + ln <= 32 and off <= mx, (m, k, seeds, p, subs)
                        assert off == win - (a0 + l0 - a) - k, (m, k, seeds, p, (a, ln, off), win)
                        last = a + ln
                # a piece goes untested only when fewer than k + 1 rows lie around it
                for p, (a0, l0) in enumerate(seeds):
                    assert (p in rows) or (a0 + (m - a0 - l0) < k + 1) or win == 0, (m, k, seeds, p)
    with pytest.raises(sassy.SassyHipError):
        sassy.seed_test_rows(40, 2, [(0, 6), (6, 7), (13, 7)])


def test_sub_piece_test_never_rejects_a_match(sassy):
    """Soundness of the sub-piece test's layout, on the CPU: a text that holds the pattern with at most k edits, none of
    them in seed p, is a hit of seed p that the test must pass -- replay the kernel's comparison (seed_kernels.hip:
    test_finish: window from win_left in front of the seed's end, sub-piece u at offset off + 0 .. 2k) on the exported rows
    for every shape, every tested piece, random edits (next to the seed, at sub-piece borders, insertions and deletions that
    shift what lies beyond them)."""
    import random
    rng = random.Random(20260929)
    for m in range(8, 33):
        for k in range(0, 8):
            if m // (k + 1) < 3:
                continue
            seeds = sassy.seed_layout("dna", [b"A" * m], k)
            win, mx, rows = sassy.seed_test_rows(m, k, seeds)
            for p, subs in rows.items():
                a0, l0 = seeds[p]
                for _ in range(40):
                    pat = [rng.choice("ACGT") for _ in range(m)]
                    # up to k edits at rows outside the seed; the text is built row by row
                    edits = {}
                    for _e in range(rng.randrange(0, k + 1)):
                        r = rng.choice([x for x in range(m) if not (a0 <= x < a0 + l0)])
                        edits[r] = rng.choice("sid")
                    left = [rng.choice("ACGT") for _ in range(80)]
                    text, seed_end = list(left), None
                    for r in range(m):
                        e = edits.get(r)
                        if e == "d":
                            pass
                        elif e == "s":
                            text.append(rng.choice([c for c in "ACGT" if c != pat[r]]))
                        elif e == "i":
                            text.append(rng.choice("ACGT")); text.append(pat[r])
                        else:
                            text.append(pat[r])
                        if r == a0 + l0 - 1:
                            seed_end = len(text)
                    text += [rng.choice("ACGT") for _ in range(80)]
                    cl = seed_end - win
                    ok = any(text[cl + off + d: cl + off + d + ln] == pat[a:a + ln]
                             for a, ln, off in subs for d in range(2 * k + 1))
                    assert ok, (m, k, p, seeds, subs, "".join(pat), edits)


def test_no_device_fails_loudly(sassy):
    assert sassy.device_count() == 0
    s = sassy.Searcher("dna", rc=False)
    with pytest.raises(sassy.SassyHipError, match="no usable HIP device"):
        s.search(b"ACGT", b"ACGTACGT", 0)
    e = s.encode_patterns([b"ACGT"])
    with pytest.raises(sassy.SassyHipError, match="no usable HIP device"):
        s.search_encoded_patterns(e, b"ACGTACGT", 0)
    # invalid IUPAC pattern is rejected before any device work (reference panics: iupac.rs:19-24)
    with pytest.raises(sassy.SassyHipError, match="not valid IUPAC"):
        sassy.Searcher("iupac", rc=False).search(b"AC1T", b"ACGT", 0)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_drop_in_search_aborts_without_device(sassy):
    """The drop-in `search` mirrors the reference's panic: message + abort, never a fake result."""
    code = (
        "import ctypes as C, sassy_amd\n"
        "L = sassy_amd.lib()\n"
        "s = L.sassy_searcher(b'dna', False, float('nan'))\n"
        "out = C.POINTER(sassy_amd.CMatch)()\n"
        "L.search(s, b'ACGT', 4, b'ACGTACGT', 8, 0, C.byref(out))\n"
        "print('returned')\n"
    )
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert p.returncode != 0
    assert "returned" not in p.stdout
    assert "no usable HIP device" in p.stderr


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sassy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "liboracle" not in src, f
                assert "orc_" not in src.replace("orc_generate_dna", "").replace("orc_make_plant", "").replace("orc_plant_window", ""), f


def test_cli_fastx_reader(tmp_path):
    """FASTA (multi-line), FASTQ and gzip input of the CLI front end (host logic, no GPU)."""
    import gzip
    from sassy_amd.cli import read_fastx
    (tmp_path / "a.fa").write_text(">chr1 desc\nACGT\nACGT\n>chr2\nTTTT\n")
    (tmp_path / "b.fq").write_text("@r1 x\nACGTN\n+\nIIIII\n@r2\nGG\n+\n##\n")
    with gzip.open(tmp_path / "c.fa.gz", "wb") as fh:
        fh.write(b">z\nAC\nGT\n")
    assert list(read_fastx(str(tmp_path / "a.fa"))) == [("chr1 desc", b"ACGTACGT"), ("chr2", b"TTTT")]
    assert list(read_fastx(str(tmp_path / "b.fq"))) == [("r1 x", b"ACGTN"), ("r2", b"GG")]
    assert list(read_fastx(str(tmp_path / "c.fa.gz"))) == [("z", b"ACGT")]


def test_c_client_compiles_and_links_against_the_headers(tmp_path):
    """include/sassy.h is a C header and libsassy_hip.so a C library: a C11 translation unit that uses
    the reference's four symbols compiles without warnings and links with -lsassy_hip (it is RUN,
    against the oracle, by the GPU suite: test_compiled_c_client_links_and_runs)."""
    exe = str(tmp_path / "dropin_client")
    p = subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c11", os.path.join(ROOT, "tests", "c", "dropin_client.c"),
                        "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "sassy_amd", "lib"), "-lsassy_hip",
                        "-lpthread", "-lm", "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    # the additive header is plain C as well
    src = tmp_path / "hdr.c"
    src.write_text('#include "sassy_hip.h"\nint main(void) { return sassy_hip_required_halo(32, 3) ? 0 : 1; }\n')
    p = subprocess.run(["gcc", "-Wall", "-Wextra", "-Werror", "-std=c11", str(src), "-I" + os.path.join(ROOT, "include"),
                        "-L" + os.path.join(ROOT, "sassy_amd", "lib"), "-lsassy_hip", "-o", str(tmp_path / "hdr")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "sassy_amd", "lib")
    assert subprocess.run([str(tmp_path / "hdr")], env=env).returncode == 0


def test_rust_shim_declares_only_exported_symbols(sassy):
    """bindings/rust/src/lib.rs is source only (no rustc in this image); at least every extern "C" function it
    declares must be a symbol libsassy_hip.so exports and include/sassy_hip.h (or sassy.h) declares."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    names = re.findall(r"fn\s+(sassy_\w+)\s*\(", block)
    assert len(names) >= 12
    hdr = open(os.path.join(ROOT, "include", "sassy_hip.h")).read() + open(os.path.join(ROOT, "include", "sassy.h")).read()
    L = sassy.lib()
    for n in names:
        assert hasattr(L, n), n
        assert re.search(r"\b" + n + r"\s*\(", hdr), n


def test_switch_table_is_the_only_reader_of_the_environment(sassy):
    """csrc/switches.h: one table of switches, read once when a searcher is made (defaults, then SASSY_HIP_<NAME>), changed
    per searcher by sassy_hip_set_option -- no getenv on any search entry point.  Every switch is forced by some GPU test
    (its environment name or a set_option call appears in tests/), and DESIGN.md lists it."""
    rows = sassy.option_table()
    names = [r[0] for r in rows]
    assert len(names) == len(set(names)) >= 40 and "fused" in names and "devices" in names
    # one reader of the environment in the whole library
    n_getenv = 0
    for f in os.listdir(os.path.join(ROOT, "sassy_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            n_getenv += open(os.path.join(ROOT, "sassy_amd", "csrc", f)).read().count("getenv(")
    assert n_getenv <= 2, n_getenv   # (load_switches: one call; one mention in a comment)
    # the environment is read when the searcher is made; set_option / get_option work per searcher (no device needed)
    code = ("import os, sassy_amd\n"
            "os.environ['SASSY_HIP_FILTER_KIND'] = '4'\n"
            "a = sassy_amd.Searcher('dna', rc=False)\n"
            "os.environ['SASSY_HIP_FILTER_KIND'] = '3'\n"
            "b = sassy_amd.Searcher('dna', rc=False)\n"
            "assert (a.get_option('filter_kind'), b.get_option('filter_kind')) == (4, 3)\n"
            "a.set_option('filter_kind', 1); a.set_fused(False)\n"
            "assert (a.get_option('filter_kind'), a.get_option('fused'), b.get_option('fused')) == (1, 0, 1)\n"
            "try:\n    a.set_option('no_such_switch', 1)\nexcept sassy_amd.SassyHipError as e:\n    assert 'no such option' in str(e)\nelse:\n    raise SystemExit(2)\n"
            "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("SASSY_HIP_")}
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout + p.stderr
    # every switch has a forced test and a line in DESIGN.md
    tests_src = "".join(open(os.path.join(ROOT, "tests", f)).read() for f in os.listdir(os.path.join(ROOT, "tests")) if f.endswith(".py"))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    for n in names:
        forced = ("SASSY_HIP_" + n.upper()) in tests_src or f'set_option("{n}"' in tests_src
        assert forced, f"switch {n} is forced by no test"
        assert f"`{n}`" in design, f"switch {n} is missing from DESIGN.md's table"


def test_fastx_batches_equal_a_line_by_line_parse(tmp_path):
    """sassy_amd/fastx.py (numpy over an mmap: single-line records without a copy, wrapped FASTA by strided copies, .gz by a
    zlib stream) against a plain line-by-line parse: wrapped / unwrapped / ragged FASTA, CRLF, empty records, FASTQ, gzip,
    batches smaller than a record, a file without a final newline."""
    import gzip
    import random
    from sassy_amd.fastx import read_fastx_batches

    def slow(path):
        data = gzip.open(path, "rb").read() if open(path, "rb").read(2) == b"\x1f\x8b" else open(path, "rb").read()
        lines = data.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()
        lines = [ln.rstrip(b"\r") for ln in lines]
        out = []
        if lines and lines[0].startswith(b"@"):
            for i in range(0, len(lines), 4):
                out.append((lines[i][1:].decode(), lines[i + 1]))
            return out
        rid, seq = None, []
        for ln in lines:
            if ln.startswith(b">"):
                if rid is not None:
                    out.append((rid, b"".join(seq)))
                rid, seq = ln[1:].decode(), []
            else:
                seq.append(ln)
        if rid is not None:
            out.append((rid, b"".join(seq)))
        return out

    rng = random.Random(5)
    seq = lambda n: bytes(rng.choice(b"ACGTN") for _ in range(n))
    files = {}
    wrap = lambda s, w, eol: eol.join(s[i:i + w] for i in range(0, len(s), w))
    files["wrapped.fa"] = b"".join(b">chr%d some text\n" % i + wrap(seq(n), 60, b"\n") + b"\n" for i, n in enumerate([1000, 60, 61, 0, 5, 7000, 120]))
    files["crlf.fa"] = b"".join(b">r%d\r\n" % i + wrap(seq(n), 70, b"\r\n") + b"\r\n" for i, n in enumerate([300, 70, 1]))
    files["ragged.fa"] = b">a\nACGT\nAC\nACGTAC\n>b\nAAAA\nCCCC\nGG\n>c\n>d\nT\n"
    files["unwrapped.fa"] = b"".join(b">read%d\n" % i + seq(rng.randrange(1, 200)) + b"\n" for i in range(500))
    files["no_final_newline.fa"] = b">x\nACGT\nAC"
    files["reads.fq"] = b"".join(b"@q%d extra\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in enumerate(seq(rng.randrange(1, 150)) for _ in range(400)))
    files["at_quality.fq"] = b"@r1\nACGT\n+\n@@@@\n@r2\nGG\n+\n@I\n"
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    with gzip.open(tmp_path / "wrapped.fa.gz", "wb") as fh:
        fh.write(files["wrapped.fa"])
    with gzip.open(tmp_path / "reads.fq.gz", "wb") as fh:
        fh.write(files["reads.fq"])
    for name in list(files) + ["wrapped.fa.gz", "reads.fq.gz"]:
        want = slow(str(tmp_path / name))
        for bb in (64, 1000, 1 << 20):
            got = []
            for rb in read_fastx_batches(str(tmp_path / name), bb):
                assert int(rb.texts.lens.sum()) == rb.text_bytes
                got += [(rb.id(i), rb.sequence(i)) for i in range(len(rb))]
            assert got == want, (name, bb, len(got), len(want))
