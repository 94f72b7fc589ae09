#!/bin/bash
# gpurun recipe: the region server's own tests, one at a time under a timeout (a first run of a resident kernel: nothing may be
# left spinning on the box).   usage (on the GPU box): bash tools/run/server_first.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
: > gpurun_out/server_first.log
for t in test_the_server_gives_what_the_launched_pipeline_gives test_the_server_gives_what_the_oracle_pipeline_gives \
         test_private_handles_past_four_go_through_the_server_by_default test_the_server_leaves_the_chip_when_idle_and_comes_back \
         test_calls_outside_the_servers_limits_take_the_launched_pipeline test_every_forward_geometry_of_the_server test_two_tickets_per_thread_on_the_shared_handle \
         test_a_region_gives_the_same_bits_alone_and_beside_other_callers; do
    echo "== $t" >> gpurun_out/server_first.log
    timeout 180 python -m pytest tests/test_server_hip.py -x -q -m gpu -k "$t" 2>&1 | tail -25 >> gpurun_out/server_first.log
    rc=${PIPESTATUS[0]}
    echo "rc=$rc" >> gpurun_out/server_first.log
    if [ "$rc" != "0" ]; then break; fi
done
tail -60 gpurun_out/server_first.log
