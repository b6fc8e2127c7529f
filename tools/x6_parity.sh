#!/bin/bash
# The reference-golden parity cases on the split-bf16 EXPERIMENT build: which streams stay identical, where the
# first decision flips, dPSNR -- written next to the committed fp32 floors.  Run on an MI355X from the repo root:
#   python -c "import rgbd_amd._lib as L; L.build_experiment_x6()" && bash tools/x6_parity.sh [out.json]
out=${1:-gpurun_out/x6_floors.json}
mkdir -p "$(dirname "$out")"
RGBD_AMD_LIB=$PWD/learning-based-rgb-d-image-compression_amd/librgbd_amd_x6.so RGBD_RECORD_FLOORS=$PWD/$out \
  timeout -k 10 900 python -m pytest tests/test_gpu_parity_pinned.py -q 2>&1 | tail -6
python3 - "$out" <<'PY'
import json, sys
a = json.load(open("tests/golden/parity_floors.json"))
b = json.load(open(sys.argv[1]))
keys = ("clean_parts_vs_golden", "identical_streams", "identical_z", "dlen_r", "dlen_d", "dlen", "dpsnr_r", "dpsnr_d", "dpsnr",
        "flip_kind", "flip_part", "flip_ref_margin")
def fmt(d):
    return "  ".join(f"{k.replace('clean_parts_vs_golden', 'clean_parts')}={d[k] if not isinstance(d[k], float) else '%.2g' % d[k]}" for k in keys if k in d)
for k in sorted(b):
    print(k)
    print("   fp32 MFMA:", fmt(a.get(k, {})))
    print("   bf16 x 6 :", fmt(b[k]))
print("identical streams: fp32 %d, bf16 x 6 %d of %d cases" % (sum(bool(a[k].get("identical_streams")) for k in b if k in a),
                                                               sum(bool(b[k].get("identical_streams")) for k in b), len(b)))
PY
