#!/bin/bash
# gfx950 assembly of one kernel source with the flags of the shipped build: tools/asm_of.sh tmpnn_edge_wave.hip [extra flags] > x.s
cd "$(dirname "$0")/../thermompnn_amd/csrc"
f=$1; shift
nan3="-mno-amdgpu-ieee -fno-honor-nans"
case $f in tmpnn_split.hip|tmpnn_edge.hip|tmpnn_msg.hip|tmpnn_edge_msg.hip|tmpnn_edge_wave.hip|tmpnn_node.hip) nan3="-DTM_GELU_NAN3=1";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-inline-asm -mcode-object-version=5 $nan3 "$@" --offload-device-only -S "$f" -o -
