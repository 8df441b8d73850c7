for lib in diffusionremotesensing_amd/libdrs_hip.so libdrs_r4b2.so libdrs_r2b4.so libdrs_r1b4.so; do
DRS_LIB=$PWD/$lib DRS_BENCH_OPS=gpurun_out/ab_ops.txt python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
python -c "
import json
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$lib', d['value'])
"
grep conv0 gpurun_out/ab_ops.txt | cut -c1-60
done
DRS_LIB=$PWD/libdrs_r2b4.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward or sample" 2>&1 | tail -2
