export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_generic.py tests/test_gpu_config4_ref.py -m gpu -x -q 2>&1 | tail -3 || exit 1
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/kt -o run -- python3 tools/config4_bench.py > /dev/null 2>&1
python3 tools/c4_timeline.py $(find gpurun_out/kt -name '*kernel_trace.csv' | head -1)
grep -E "pack_rows|to_bf16" $(find gpurun_out/kt -name '*kernel_stats.csv' | head -1) | cut -d, -f1-4 | cut -c1-40,150-
