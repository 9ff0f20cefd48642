#!/bin/bash
# builds one binary per variant into tools/exp_gram/bin (shipped to the GPU box with the snapshot, git-ignored)
cd "$(dirname "$0")"; mkdir -p bin
b() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT_NAME="\"$name\"" "$@" -o bin/$name harness.hip & }
b base -DSRC_INC='"gram_c3.inc"'
b lb3 -DSRC_INC='"gram_c3.inc"' -DLB=3
b lb4 -DSRC_INC='"gram_c3.inc"' -DLB=4
b nostore -DSRC_INC='"gram_c3.inc"' '-DSTORE_COND=(flags==12345)'
b nomath -DSRC_INC='"gram_c3.inc"' -DNOMATH=1
b nomath_lb4 -DSRC_INC='"gram_c3.inc"' -DNOMATH=1 -DLB=4
b v2 -DSRC_INC='"gram_c3_v2.inc"' -DSMAX=8
b v2_lb3 -DSRC_INC='"gram_c3_v2.inc"' -DSMAX=8 -DLB=3
b v2_nostore -DSRC_INC='"gram_c3_v2.inc"' -DSMAX=8 '-DSTORE_COND=(flags==12345)'
b v2_nomath -DSRC_INC='"gram_c3_v2.inc"' -DSMAX=8 -DNOMATH=1
b v3 -DSRC_INC='"gram_c3_v3.inc"' -DSMAX=8
b v3_lb3 -DSRC_INC='"gram_c3_v3.inc"' -DSMAX=8 -DLB=3
b v3_nomath -DSRC_INC='"gram_c3_v3.inc"' -DSMAX=8 -DNOMATH=1
b v4 -DSRC_INC='"gram_c3_v4.inc"'
b v4_lb3 -DSRC_INC='"gram_c3_v4.inc"' -DLB=3
b v4_nostore -DSRC_INC='"gram_c3_v4.inc"' '-DSTORE_COND=(flags==12345)'
wait
ls -la bin
