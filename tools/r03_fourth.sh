#!/bin/bash
# round 3, fourth GPU call: (1) does de-phasing the two workgroups of a CU relieve the HBM-bound backward sweeps?
# (development build, ISDF_DEBUG_STAGGER_GEN=1: second-round workgroups start k kilo-cycles late); (2) host profile of step()
O=gpurun_out/r03d; mkdir -p $O
for op in fp16 fp16x2; do
for k in 0 20 40 60 80 100; do
  echo "== $op stagger $k kcycles" >> $O/stagger.txt
  ISDF_FWD_OPERAND=$op ISDF_DEBUG_STAGGER_GEN=1 ISDF_DEBUG_STAGGER=$k TIMELINE_BRIEF=1 python tools/timeline.py 2>&1 | grep -E "SUMMARY|workgroups sampled" >> $O/stagger.txt
done
done
cat $O/stagger.txt
python tools/profile_step_host.py > $O/host_profile.txt 2>&1
head -50 $O/host_profile.txt
