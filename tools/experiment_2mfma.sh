#!/bin/bash
# Verdict r02 item 6 (measured, NOT shipped): two MFMAs per product instead of three (the w_lo * a_hi term dropped in every
# kernel that goes through PolicyBF16X3::mma; resblock0 / the wide gate issue their products explicitly and stay exact).
# Build: tools/build_variant.sh mfma2 "-DDRS_EXPERIMENT_2MFMA=1"; run on the GPU box from the repo root.
for lib in diffusionremotesensing_amd/libdrs_hip.so variants/libdrs_mfma2.so; do
  DRS_LIB=$PWD/$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/x2.json 2> gpurun_out/x2.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/x2.json').read().strip().splitlines()[-1])
print(json.dumps({"lib": "$lib", "steps_per_s": d["value"], "psnr_vs_ref_db": d.get("psnr_vs_ref_db"), "psnr_detail": d.get("psnr_detail"),
                  "family_tflops": d["roofline"]["achieved"]}))
PY
done
