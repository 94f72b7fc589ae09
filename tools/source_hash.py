"""The hash of the kernel sources a kernel family of lorikeet_amd/libphmm.so is compiled from: what a committed PMC entry of
profiles/pmc_traffic.json is keyed on (bench.py), and what the library carries from its own build (phmm_build_info(): the
Makefile bakes the output of this script into phmm_build_info.o, so a library built from other sources than the tree's says so).
usage: python tools/source_hash.py [family ...]   ->  family=hash, one per line (all families when none is named)"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = {  # what each kernel family is compiled from (lorikeet_amd/csrc)
    "pairhmm": ("phmm_device.hpp", "phmm_internal.hpp", "phmm_kernels.hip", "phmm_chain_kernels.hip", "phmm_chain32_kernels.hip",
                "phmm_exact_kernels.hip", "phmm_engine_kernels.hip", "phmm_prep_device.hpp", "phmm_post_device.hpp"),
    "sw": ("phmm_sw_internal.hpp", "phmm_sw_kernels.hip", "phmm_sw_device.hpp"),
    "cigar": ("phmm_cigar_internal.hpp", "phmm_cigar_kernels.hip", "phmm_cigar_device.hpp"),
    "server": ("phmm_server.hpp", "phmm_server_kernels.hip"),
}


def source_hash(family="pairhmm"):
    h = hashlib.sha256()
    for name in sorted(KERNEL_SOURCES[family]):
        h.update(name.encode())
        h.update(open(os.path.join(ROOT, "lorikeet_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def build_info():
    """The string phmm_build_info() returns for a library built from this tree."""
    return " ".join("%s=%s" % (f, source_hash(f)) for f in sorted(KERNEL_SOURCES))


if __name__ == "__main__":
    fams = sys.argv[1:] or sorted(KERNEL_SOURCES)
    if fams == ["--build-info"]:
        print(build_info())
    else:
        for f in fams:
            print("%s=%s" % (f, source_hash(f)))
