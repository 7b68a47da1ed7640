# A/B of the persistent k_shade_lean grid: AWSM_LEAN_WGS_PER_CU x library variant, overlapped frames (bench.py default)
mkdir -p gpurun_out/r2i; : > gpurun_out/r2i/ab.log
run() {  # lib per_cu extra...
  lib=$1; n=$2; shift 2
  echo "== $lib per_cu $n $*" >> gpurun_out/r2i/ab.log
  AWSM_HIP_LIB=$lib AWSM_LEAN_WGS_PER_CU=$n timeout -k 10 200 python bench.py --allow-variant-lib --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline'].get('all_kernels_ms'))" >> gpurun_out/r2i/ab.log
}
L=awsm-renderer_amd/libawsm_hip.so
for n in ${GRID_LIST:-0 4}; do run $L $n; done
for v in ${VARIANTS}; do for n in ${VGRID:-4}; do run build/variants/lib_$v.so $n; done; done
run $L 0 --no-overlap
if [ -z "$NO_TESTS" ]; then timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not rehearsal" > gpurun_out/r2i/pytest.log 2>&1; tail -3 gpurun_out/r2i/pytest.log; fi
bash tools/frame_timeline.sh gpurun_out/r2i/tl > gpurun_out/r2i/tl.txt 2>&1; cat gpurun_out/r2i/tl.txt
cat gpurun_out/r2i/ab.log
