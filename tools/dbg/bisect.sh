#!/bin/bash
cd "$(dirname "$0")/../.."
T=tests/test_region_hip.py::test_two_callers_with_private_handles_align_every_pair_at_the_same_time
for f in test_abi_hardening test_calculate_cigar_hip test_cpp_host_layer test_engine_hip test_f32_first test_golden_repetitions test_hip_fuzz test_hip_parity test_multi test_project_hip test_realign_hip test_region_handoffs; do
  r=$(python -m pytest tests/$f.py $T -q -m gpu --deselect tests/test_region_handoffs.py::test_cpp_callers_verify_every_call_against_its_first_pass 2>&1 | tail -1)
  echo "$f: $r"
done
