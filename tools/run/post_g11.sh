# after tools/run/g11.sh came back: rocpd databases -> profiles/*_summary.txt, *_pmc.json, pmc_traffic.json
cd /root/repo
for T in config2_f64 config2_f32 config3 config5 ragged; do
  python tools/rocpd_summary.py gpurun_out/r02_$T profiles/r02_$T > /dev/null
  python tools/pmc_update.py gpurun_out/r02_$T profiles/r02_$T > /dev/null
done
for T in engine_call sw_bench; do
  python tools/rocpd_summary.py gpurun_out/r02_$T profiles/r02_$T > /dev/null
  grep -v amdgpu gpurun_out/r02_$T/bench.txt >> profiles/r02_${T}_summary.txt
done
python tools/pmc_update_sw.py gpurun_out/r02_sw_bench profiles/r02_sw_bench > /dev/null
python - <<'PY'
import json
for e in json.load(open("profiles/pmc_traffic.json")):
    print(e["workload"], e["regions"], e["precision"], e["kernel_short"], e["src_hash"], "hbm %.3g" % e["hbm_bytes_per_launch"], "valu %.4g" % e["valu_insts_per_launch"])
PY
