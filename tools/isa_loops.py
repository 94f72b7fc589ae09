"""Developer tool: the opcode table of a kernel's inner loops, from the ISA hipcc emits for gfx950 (no GPU needed).

usage: python tools/isa_loops.py [--min N] [--asm out.s] <source.hip> <regex on the demangled kernel name> [hipcc flags ...]
  e.g. python tools/isa_loops.py lorikeet_amd/csrc/phmm_chain_kernels.hip 'phmm_forward_chain_k<16, 19>' -DPHMM_CHAIN_L=16
For every backward branch whose body is longer than --min instructions (default 100): instruction counts by class (VALU by
opcode, SALU, LDS, VMEM, waits/nops), the kernel's register and scratch figures, and the VALU count of the loop -- what
`SQ_INSTS_VALU / steps` of a PMC pass has to agree with.  The tables of NOTEBOOK.md were made with it."""
import collections
import os
import re
import subprocess
import sys
import tempfile

args = sys.argv[1:]
min_len = 100
if "--min" in args:
    i = args.index("--min")
    min_len = int(args[i + 1])
    del args[i:i + 2]
asm_out = None
if "--asm" in args:   # also write the kernel's ISA (comment lines dropped) to this file
    i = args.index("--asm")
    asm_out = args[i + 1]
    del args[i:i + 2]
src, pattern, flags = os.path.abspath(args[0]), re.compile(args[1]), args[2:]
tmp = tempfile.mkdtemp(prefix="isa_")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-save-temps", "-c", src, "-o", "k.o"] + flags,
                      cwd=tmp, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
lines = open(os.path.join(tmp, asm)).read().split("\n")
names = [l.split(":")[0] for l in lines if re.match(r"^_Z\w+:", l)]
filt = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for mangled, nice in zip(names, filt):
    if not pattern.search(nice):
        continue
    a = next(i for i, l in enumerate(lines) if l.startswith(mangled + ":"))
    b = next(i for i in range(a, len(lines)) if ".amdhsa_kernel" in lines[i])
    end = next(i for i in range(b, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    meta = {k: re.search(k + r"\s+(\d+)", "\n".join(lines[b:end])) for k in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size", "accum_offset")}
    if asm_out:
        open(asm_out, "w").write("\n".join(l for l in lines[a:b] if not l.lstrip().startswith(";")) + "\n")
    body = [l.split(";")[0].strip() for l in lines[a:b]]
    body = [l for l in body if l and not l.startswith(".") or re.match(r"^\.LBB\d+_\d+:", l or "")]
    print(f"== {nice}\n   vgpr+agpr {meta['next_free_vgpr'].group(1)}, sgpr {meta['next_free_sgpr'].group(1)}, scratch "
          f"{meta['private_segment_fixed_size'].group(1)} B, instructions {sum(1 for l in body if not l.endswith(':'))}")
    label_at = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    back = []          # (first, last) instruction index of every backward branch's body
    for i, l in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            back.append((label_at[m.group(1)], i, m.group(1)))

    def table(seg):
        ops = collections.Counter()
        for x in seg:
            o = x.split()[0]
            ops[o + ("(dpp)" if re.search(r"row_shr|wave_shr|row_bcast|quad_perm", x) else "")] += 1
        cls = collections.Counter()
        for o, c in ops.items():
            cls["VALU" if o.startswith("v_") else "LDS" if o.startswith("ds_") else "VMEM" if re.match(r"(global|buffer|scratch|flat)_", o)
                else "wait/nop" if o in ("s_waitcnt", "s_nop", "s_barrier") else "SALU/branch"] += c
        return (", ".join(f"{k} {v}" for k, v in cls.most_common()),
                "; ".join(f"{o} {c}" for o, c in ops.most_common()))

    for lo, hi, name in back:
        if hi - lo < min_len or any(l2 >= lo and h2 <= hi and (l2, h2) != (lo, hi) and h2 - l2 >= min_len for l2, h2, _ in back):
            continue   # too short, or it contains another loop that is listed on its own
        loop = [x for x in body[lo:hi + 1] if not x.endswith(":")]
        print(f"-- loop {name}: {len(loop)} instructions; {table(loop)[0]}")
        # straight-line pieces between branches / labels: the big ones are the steps, the rest runs once per read or never
        seg = []
        for x in body[lo:hi + 1] + ["<end>:"]:
            if x.endswith(":") or re.match(r"s_c?branch", x):
                if not x.endswith(":"):
                    seg.append(x)
                if len(seg) >= 40:
                    c, o = table(seg)
                    print(f"   piece of {len(seg)}: {c}\n      {o}")
                seg = []
            else:
                seg.append(x)
