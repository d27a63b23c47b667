#!/bin/bash
# Builds the committed (HEAD) kernels as libshpair_prev.so next to the working-tree build,
# for an interleaved A/B with tools/ab_libs.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/shpair_head && mkdir -p /tmp/shpair_head
git archive HEAD lammps-spherharm_amd/csrc include | tar -x -C /tmp/shpair_head
make -s -j8 -C /tmp/shpair_head/lammps-spherharm_amd/csrc OUT="$PWD/lammps-spherharm_amd/shpair/libshpair_prev.so" >/dev/null
make -s -j8 -C lammps-spherharm_amd/csrc >/dev/null
ls -la lammps-spherharm_amd/shpair/*.so
