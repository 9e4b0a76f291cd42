set -e
mkdir -p gpurun_out
: > gpurun_out/r3_emu.log
for W in 1 2 4 8; do
  SP=780; [ $W -ge 8 ] && SP=600
  for S in 0 $SP; do
    [ $W -eq 1 ] && [ $S -ne 0 ] && continue
    echo "== world $W split $S" >> gpurun_out/r3_emu.log
    timeout -k 10 300 python tools/rank_emulation.py --world $W --split $S --steps 10 >> gpurun_out/r3_emu.log 2>&1
  done
done
tail -5 gpurun_out/r3_emu.log
