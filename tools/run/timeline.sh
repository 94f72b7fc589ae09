#!/bin/bash
# gpurun recipe: kernel timeline of one phmm_region_compute call per region from one thread, on the config-2 shape and on
# 30 x 3 regions, the all-pairs aligner on (default) and off (the chain) -> profiles/<round>_region_timeline.txt
# usage (from the repository root, on the GPU box): bash tools/run/timeline.sh r04
set -u
round=${1:-r04}
export TMPDIR=/tmp
out=gpurun_out/${round}_region_timeline.txt
mkdir -p gpurun_out
echo "# one phmm_region_compute call per region from one thread (tools/threads_bench TB_MODE=fused), rocprofv3 --kernel-trace; tools/region_timeline.py" > $out
for shape in "128 8" "30 3"; do
    set -- $shape
    for all in -1 0; do
        d=/tmp/tl_$1_$2_$all
        rm -rf $d
        echo "## $1 $2 reads x haplotypes, R=150, H=300; PHMM_REGION_SW_ALL=$all" >> $out
        PHMM_REGION_SW_ALL=$all TB_MODE=fused TB_THREADS=1 rocprofv3 --kernel-trace --output-format csv -d $d -- tools/threads_bench 0.3 $1 $2 150 300 1 2>/dev/null | tail -1 >> $out
        python tools/region_timeline.py $d | head -8 >> $out
    done
done
cat $out
