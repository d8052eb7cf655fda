for w in S3 S2; do for l in 2 4 8 16; do
 ADMP_PAIR_LPR=$l ADMP_FIELD_LPR=$l python tools/kernels.py $w 5 > gpurun_out/sw_${w}_$l.log 2>&1 || exit 1
 echo "$w lpr=$l static: $(grep -A1 "$w static" gpurun_out/sw_${w}_$l.log | grep -o "pair_f[a-z_]* [0-9.]*" | tr '\n' ' ') moving: $(grep -A1 "$w moving" gpurun_out/sw_${w}_$l.log | grep -o "pair_f[a-z_]* [0-9.]*" | tr '\n' ' ')"
done; done
