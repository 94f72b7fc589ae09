#!/bin/bash
# Re-embeds integration/hip_ffi.rs and integration/hip_backend.rs into integration/lorikeet-hip.patch (the patch adds them
# as src/pair_hmm/hip_ffi.rs and src/pair_hmm/hip_backend.rs).
# Needs the reference tree (build container only): applies the current patch to a scratch copy of the files it touches,
# replaces the FFI file, and diffs again.  usage: integration/refresh_patch.sh [/root/reference]
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
W=$(mktemp -d)
cd "$W" && git init -q .
for f in $(grep '^diff --git' "$HERE/lorikeet-hip.patch" | sed 's#diff --git a/\(\S*\) .*#\1#'); do
  if [ -f "$REF/$f" ]; then mkdir -p "$(dirname "$f")"; cp "$REF/$f" "$f"; fi
done
git add -A && git -c user.email=a@b -c user.name=x commit -q -m base
git apply "$HERE/lorikeet-hip.patch"
cp "$HERE/hip_ffi.rs" src/pair_hmm/hip_ffi.rs
cp "$HERE/hip_backend.rs" src/pair_hmm/hip_backend.rs
git add -A && git diff --cached > "$HERE/lorikeet-hip.patch"
echo "refreshed: $(grep -c '^+' "$HERE/lorikeet-hip.patch") added lines"
rm -rf "$W"
