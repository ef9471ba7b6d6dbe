# A/B timing of library variants on ONE box (box-to-box spread is +-3 %):  bash tools/ab.sh libA.so libB.so ...
# (variants live in bhnerf_amd/csrc/; default: libbhnerf_hip_A.so against the product build)
LIBS=${@:-libbhnerf_hip_A.so libbhnerf_hip.so}
for r in 1 2; do for l in $LIBS; do echo -n "$l "; BHNERF_HIP_LIB=$PWD/bhnerf_amd/csrc/$l python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-tutorial-domain --no-parity-mode --no-other-configs --no-width128 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms']; print(round(d['ms_per_step'],3), [round(v,3) for v in k.values()])"; done; done
