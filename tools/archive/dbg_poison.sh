# which forest allocation is read before it is written?  (AZG_DEBUG_POISON, csrc/azg.hip dalloc): runs tools/dbg_cadence2.py with
# every allocation poisoned in turn; a line whose example counts differ from the clean run names the buffer
for m in 0 0x7ffcfe 0x2 0x4 0x8 0x20 0x80 0xfc00 0x3f0000; do
  echo "poison mask $m: $(AZG_DEBUG_POISON=$m PYTHONPATH=tests:. timeout 300 python tools/dbg_cadence2.py 1 2>&1 | tail -1)"
done
