#!/bin/bash
# Developer tool: compile the kernels and print VGPR/AGPR/occupancy per instantiation (filter: $1)
cd /tmp && mkdir -p st && cd st
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPHMM_L=${PHMM_L:-16} -DPHMM_WITH_GENERIC -save-temps -c /root/repo/lorikeet_amd/csrc/phmm_kernels.hip -o k.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|AGPRs|Occupancy" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - | sed 's/Function Name: _ZN4phmm//; s/EEEvNS_13ForwardParamsE//; s/12phmm_forwardILi/fwd</; s/ELi/,/' | grep -E "${1:-.}"
