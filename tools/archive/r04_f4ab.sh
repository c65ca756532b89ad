cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04p
python -m pytest tests/test_gpu_mcts.py -q -m gpu -k "hashnet" 2>&1 | grep -E "^FAILED|^E |passed|failed" | head -8
{
echo '# f4 game plugins on one MI355X (`tools/bench_f4.py --md [--net hashhip | mlp]`: batched self-play, 200 simulations per move, 20 timed ply waves after 2 of warm-up, HIP-graph rounds)'
echo
echo 'Evaluators: `hash` = the integer hash-net of the parity tests as ~35 torch ops per round (the round-3 figure), `hashhip` = the same function as ONE engine kernel (`azg_eval_hashnet`, bit-identical: `test_device_hashnet_equals_torch_hashnet`) -- i.e. the tree + env side of a plugin without evaluator launch overhead, `mlp` = a PyTorch module of the size of the reference small nets through `nnet.TorchModuleEvaluator`.'
echo
echo '| game | players | state B | actions | games | evaluator | plies/s | M sims/s | ms / round | levels / sim | valid / level | errors | validate | forest GB |'
echo '|---|---|---|---|---|---|---|---|---|---|---|---|---|---|'
python tools/bench_f4.py --md --plies 20 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net hashhip 2>/dev/null
python tools/bench_f4.py --md --plies 20 --net mlp 2>/dev/null
} > gpurun_out/r04p/f4_bench.md
cat gpurun_out/r04p/f4_bench.md
