# profiles/r06_f4_bench.md: the six f4 plugins with their evaluators (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06f4; mkdir -p $O
{
echo '# f4 game plugins on one MI355X (`tools/bench_f4.py --md --net <evaluator>`: batched self-play, 200 simulations per move, 20 timed ply waves after 2 of warm-up, HIP-graph rounds)'
echo
echo 'Evaluators: `engine` = the game'"'"'s SHIPPED net (minivilles/pretrained_2players.pt nn_version 82, thelittleprince/pretrained_3players.pt nn_version 83: both of the MobileNet-1d family) as ONE launch of the engine kernel (`nn_mb1d.hip.h`, geometries `AZG_NET_MINIVILLES2` / `AZG_NET_TLP3`; golden forward vectors from the reference modules: `tests/test_nnet.py`); `torchnet` = the same weights as PyTorch-ROCm ops (`nnet.MobileNet1d`); `hashhip` = the integer hash-net of the parity tests as one engine kernel (`azg_eval_hashnet`, `include/azg_testaids.h`) -- the tree + env side of a plugin without evaluator cost; `hash` = the same function as ~35 torch ops; `mlp` = a PyTorch module of the size of the reference small nets through `nnet.TorchModuleEvaluator`.'
echo
echo '| game | players | state B | actions | games | evaluator | plies/s | M sims/s | ms / round | levels / sim | valid / level | errors | validate | forest GB |'
echo '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|'
python tools/bench_f4.py --md --plies 20 --net engine 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net torchnet 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net hashhip 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net hash 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net mlp 2>/dev/null
python tools/bench_f4.py --md --plies 10 --net hashhip --only smallworld --games 4096 2>/dev/null | sed 's/| 4096 |/| **4096** |/'
python tools/bench_f4.py --md --plies 10 --net hashhip --only botanik --games 4096 2>/dev/null | sed 's/| 4096 |/| **4096** |/'
python tools/bench_f4.py --md --plies 10 --net engine --only minivilles --games 4096 2>/dev/null | sed 's/| 4096 |/| **4096** |/'
echo
echo '## `k_select` at 1024 trees (rocprofv3 --kernel-trace, `tools/f4prof.sh <game>`; evaluator hashhip)'
echo
for g in smallworld botanik minivilles thelittleprince abalone akropolis; do
  bash tools/f4prof.sh $g > /dev/null 2>&1
  grep "k_select<" gpurun_out/r06p/f4prof_$g.txt | head -1 | awk -F'|' -v g=$g '{printf "* %s: `k_select` %s us average over %s launches (min %s, max %s); VGPRs %s, scratch %s B\n", g, $5, $3, $6, $7, $9, $12}'
done
echo
echo '(Akropolis runs without policy-target pruning: with ~250 valid placements and 200 simulations the pruned counts are all <= 1, which the engine reports as error bit 64 -- the reference divides 0 / 0 there.)'
} > $O/f4_bench.md
cat $O/f4_bench.md | tail -40
