cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 600 python tests/fuzz_gpu.py --seconds 200 --seed 32 --focus bytes_long > gpurun_out/r3d/fuzz_bytes.log 2>&1; tail -3 gpurun_out/r3d/fuzz_bytes.log | cut -c1-400
timeout 900 python tests/fuzz_gpu.py --seconds 500 --seed 33 > gpurun_out/r3d/fuzz_all.log 2>&1; tail -3 gpurun_out/r3d/fuzz_all.log | cut -c1-400
timeout 900 python tests/fuzz_gpu.py --seconds 200 --seed 34 --focus fused > gpurun_out/r3d/fuzz_fused2.log 2>&1; tail -3 gpurun_out/r3d/fuzz_fused2.log | cut -c1-400
