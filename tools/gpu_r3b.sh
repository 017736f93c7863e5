cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
rm -rf gpurun_out/r3c/cfg3
PROBE_M=200 PROBE_K=20 PROBE_PROFILE=iupac rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r3c/cfg3 -o t -- python tools/probe_fused.py > gpurun_out/r3c/cfg3.json 2> gpurun_out/r3c/cfg3.err
tail -1 gpurun_out/r3c/cfg3.json
python tools/timeline.py gpurun_out/r3c/cfg3 filter_count
python tools/probe_fused.py | tail -1
