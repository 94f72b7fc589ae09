#!/bin/bash
# After tools/run/profiles.sh came back: rocpd databases -> profiles/<round>_*_summary.txt, *_pmc.json, pmc_traffic.json.
# usage (build container): bash tools/run/profiles_post.sh <round>
R=${1:-r05}
cd "$(dirname "$0")/../.."
for T in config2_f64 config2_f32 config3 config5 ragged; do
  [ -d gpurun_out/${R}_$T ] || continue
  python tools/rocpd_summary.py gpurun_out/${R}_$T profiles/${R}_$T > /dev/null
  python tools/pmc_update.py gpurun_out/${R}_$T profiles/${R}_$T > /dev/null
done
for T in engine_call sw_bench; do
  [ -d gpurun_out/${R}_$T ] || continue
  python tools/rocpd_summary.py gpurun_out/${R}_$T profiles/${R}_$T > /dev/null
  grep -v amdgpu gpurun_out/${R}_$T/bench.txt >> profiles/${R}_${T}_summary.txt
done
[ -d gpurun_out/${R}_sw_bench ] && python tools/pmc_update_sw.py gpurun_out/${R}_sw_bench profiles/${R}_sw_bench > /dev/null
python - <<'PY'
import json
for e in json.load(open("profiles/pmc_traffic.json")):
    print(e["workload"], e["regions"], e["precision"], e["kernel_short"], e["src_hash"], "hbm %.3g" % e["hbm_bytes_per_launch"], "valu %.4g" % (e.get("valu_insts_per_launch") or 0))
PY
