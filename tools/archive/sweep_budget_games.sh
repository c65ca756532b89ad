# work budget x advance cadence over WHOLE games (the phase of the game changes the round, so windows must cover the same plies):
#   GAMES="splendor2 azul" CFGS="20:16 10:48" bash tools/sweep_budget_games.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in ${GAMES:-splendor2}; do for cfg in ${CFGS:-20:16 10:48}; do
  wb=${cfg%%:*}; ae=${cfg##*:}
  python bench.py --game $g --work-budget $wb --advance-every $ae --no-secondary --no-cpu-baseline --roofline-rounds 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$g whole games: wb $wb adv $ae value', round(d['value']), 'from_sims', round(d['value_from_sims']), 'ms/round', round(d['ms_per_round'],4), 'games', d['games_finished'], 'steps', d['steps'], 'err', d['engine_errors'], flush=True)"
done; done
