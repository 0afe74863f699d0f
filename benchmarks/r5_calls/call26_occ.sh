#!/bin/bash
# round 5 call 26: narrow conv kernels at capped occupancy (lab: DVD_CONV_LDS = dynamic LDS bytes per workgroup, unused by the kernel)
cd /root/repo; mkdir -p gpurun_out/r5
O=gpurun_out/r5/call26.txt
{
for lds in 0 40000 65536 131072; do
echo "== DVD_CONV_LDS=$lds (0: 3 workgroups per CU by registers; 40000: 4 -> still 3; 65536: 2; 131072: 1), 32 documents"
DVD_CONV_LDS=$lds python benchmarks/native_profile.py 32 4 --lab 2>&1 | grep "prestage\|prepare_docs\|documents per batch"
done
for lds in 0 65536 131072; do
echo "== DVD_CONV_LDS=$lds, 1 document"
DVD_CONV_LDS=$lds python benchmarks/native_profile.py 1 12 --lab 2>&1 | grep "prestage\|prepare_docs\|documents per batch"
done
} > $O 2>&1
cat $O
