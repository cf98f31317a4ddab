#!/bin/bash
# usage: build_var2.sh NAME FILE(.hip, in csrc) "EXTRA FLAGS" -> scratch/abl2/libNAME.so (only FILE is recompiled with the flags)
set -e
cd /root/repo/torch_nf_amd/csrc
NAME=$1; FILE=$2; shift; shift
B=${FILE%.hip}
mkdir -p /root/repo/scratch/abl2/obj_$NAME
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result $@ -c $FILE -o /root/repo/scratch/abl2/obj_$NAME/$B.o
OBJS=$(ls ../lib/obj/*.o | grep -v "/$B.o")
hipcc -shared --offload-arch=gfx950 -o /root/repo/scratch/abl2/lib$NAME.so $OBJS /root/repo/scratch/abl2/obj_$NAME/$B.o
echo built $NAME
