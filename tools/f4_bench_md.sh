# profiles/r03_f4_bench.md: the six f4 plugins with both evaluators (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
{
echo '# f4 game plugins on one MI355X (`tools/bench_f4.py --md [--net mlp]`: batched self-play, 200 simulations per move, 20 timed ply waves after 2 of warm-up, HIP-graph rounds)'
echo
echo 'Evaluator `hash` = the integer hash-net of the parity tests as PyTorch-ROCm ops (≈10 small launches per round); `mlp` = a PyTorch module of the size of the reference'"'"'s small per-game nets (flatten → 256 → 256 → 128, LayerNorm + SiLU, two-layer heads; random weights) through `nnet.TorchModuleEvaluator` inside the captured round.  `validate` = structural check of every tree on the host afterwards.'
echo
echo '| game | players | state B | actions | games | evaluator | plies/s | M sims/s | ms / round | levels / sim | valid / level | errors | validate | forest GB |'
echo '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|'
python tools/bench_f4.py --md --plies 20 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net mlp 2>/dev/null
echo
echo '(Akropolis runs without policy-target pruning: with ≈ 250 valid placements and 200 simulations the pruned counts are all ≤ 1, which the engine reports as error bit 64 — the reference divides 0 / 0 there.  Round 3: its district scoring runs on all lanes, 2.6 k → 4.6 k plies/s with the hash evaluator.)'
} > $O/f4_bench.md
cat $O/f4_bench.md | tail -20
