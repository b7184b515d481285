mkdir -p gpurun_out/r08c
for cfg in "0 8" "0 4" "0 2" "29 8" "30 8" "28 8" "7 4" "8 4" "26 8" "27 8" "25 8" "11 8" "2 8"; do
  set -- $cfg
  bash tools/clock_poll.sh gpurun_out/r08c/sus_$1_$2.poll ./tools/ubench/issue_mix sustain $1 $2 2.5 > gpurun_out/r08c/sus_$1_$2.txt
  cat gpurun_out/r08c/sus_$1_$2.txt
  tail -4 gpurun_out/r08c/sus_$1_$2.poll | head -2 | sed -e 's/=*//g' -e 's/GPU\[0\]//g' | awk '{print "    ", $0}' | cut -c1-230
done
