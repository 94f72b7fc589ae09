#!/bin/bash
# For a maintainer with a Rust toolchain and an MI355X: SURVEY 8 row f3's A/B -- `lorikeet call` on the repository's own
# test data with the AVX / scalar arm and with the HIP backend, same inputs, VCFs compared.  NOT run in the build container of
# this repository (no rustc, no cargo there): everything below is what INTEGRATION.md describes, in order.
#   usage: integration/ab_lorikeet_call.sh <lorikeet checkout> <this repository> [threads]
set -euo pipefail
LK=${1:?path to a checkout of rhysnewell/Lorikeet}
HERE=${2:?path to this repository}
T=${3:-10}
make -C "$HERE/lorikeet_amd/csrc" -j"$(nproc)"                       # lorikeet_amd/libphmm.so (hipcc --offload-arch=gfx950)
cd "$LK"
git apply --check "$HERE/integration/lorikeet-hip.patch" && git apply "$HERE/integration/lorikeet-hip.patch"
# optional, on top: one shared engine per device for all workers
# git apply "$HERE/integration/lorikeet-hip-shared-handle.patch"
PHMM_LIB_DIR="$HERE/lorikeet_amd" cargo build --release --features hip
export LD_LIBRARY_PATH="$HERE/lorikeet_amd:/opt/rocm/lib:${LD_LIBRARY_PATH:-}"
BIN=target/release/lorikeet
OUT=$(mktemp -d)
run() {  # <name> <extra arguments...>
    local name=$1; shift
    /usr/bin/time -v "$BIN" call -r tests/data/7seqs.fna -b tests/data/7seqs.reads_for_seq1_and_seq2.bam \
        -o "$OUT/$name" --threads "$T" "$@" 2> "$OUT/$name.time" || { tail -5 "$OUT/$name.time"; exit 1; }
    grep -E "Elapsed|Maximum resident" "$OUT/$name.time"
}
run scalar  --pairhmm-backend scalar      # the arm the oracle restates (f64, pair_hmm.rs:503-615)
run avx     --pairhmm-backend avx         # gkl (f32 first, f64 where f32 cannot hold the result)
run hip     --pairhmm-backend hip         # this repository's engine, f64
run hip-f32 --pairhmm-backend hip-f32     # ... in gkl's precision mode
# a region call that fails its hand-off checks is an error, not a silent difference
PHMM_MIRROR_CANARY=1 run hip-canary --pairhmm-backend hip
for v in avx hip hip-f32 hip-canary; do
    for f in "$OUT"/scalar/*/*.vcf*; do
        g=${f/$OUT\/scalar/$OUT\/$v}
        # records only: headers carry the command line
        if diff <(zcat -f "$f" | grep -v '^##') <(zcat -f "$g" | grep -v '^##') > "$OUT/$v.diff"; then echo "$v: $(basename "$f") identical to scalar"
        else echo "$v: $(basename "$f") DIFFERS from scalar: $(wc -l < "$OUT/$v.diff") diff lines ($OUT/$v.diff)"; fi
    done
done
echo "outputs under $OUT"
