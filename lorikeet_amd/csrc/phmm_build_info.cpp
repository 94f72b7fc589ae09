// What this library was built from: the hashes of the kernel sources, family by family, as tools/source_hash.py computed them
// when the Makefile compiled this file (-DPHMM_BUILD_INFO).  __graft_entry__.smoke() and bench.py compare it with the hashes of
// the tree they run in: a stale binary says so instead of being measured under the tree's name.
#include "../../include/phmm.h"

#ifndef PHMM_BUILD_INFO
#error "compile with -DPHMM_BUILD_INFO=\"...\" (the Makefile does: tools/source_hash.py --build-info)"
#endif

extern "C" const char *phmm_build_info(void) { return PHMM_BUILD_INFO; }
