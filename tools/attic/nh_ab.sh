#!/usr/bin/env bash
# In-session A/B of library builds on the Neo-Hookean clustered sweep (1 M-tet lattice): alternating runs, 3 rounds.
#   tools/attic/nh_ab.sh OUT lib1.so lib2.so ...
out=$1; shift
cd "$(dirname "$0")/../.."
: > "$out"
for rep in 1 2 3; do
  for lib in "$@"; do
    TETSIM_HIP_LIB=$PWD/$lib python tools/nh_time.py 55 clustered 2>&1 | grep neohookean | sed "s|^|$(basename $lib) |" >> "$out"
  done
done
sort -k1,1 -k4,4 -s "$out"
