"""Pin the C oracle against the LIVE reference on a much larger vector set than the committed fixtures.
Build-container only (needs /root/reference):
    python tools/check_oracle_vs_ref.py [--dir /tmp/azg_big]
= tools/gen_golden.py --big DIR  (reference -> vectors)  +  pytest tests/test_oracle_golden.py with AZG_GOLDEN_DIR=DIR.
gen_golden.py --big writes the six BASELINE variants (Splendor 2 / 3 / 4 players, Santorini 1 / 11, Azul); every other fixture the oracle
tests read (the six f4 games, the episode / 800- / 1600-simulation sets, ...) is taken from the committed tests/golden/ (linked into DIR), so
that the whole test file runs: the fresh big set pins the BASELINE games, the committed vectors the rest.  --out FILE keeps the tail."""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default='/tmp/azg_big')
    ap.add_argument('--out', default=None, help='write the tail of the pytest output (and what was generated) here')
    ap.add_argument('--skip-gen', action='store_true', help='DIR already holds a big set')
    a = ap.parse_args()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    if not a.skip_gen:
        subprocess.check_call([sys.executable, os.path.join(HERE, 'gen_golden.py'), '--big', a.dir], env=env)
    fresh = sorted(os.listdir(a.dir))
    committed = os.path.join(ROOT, 'tests', 'golden')
    linked = []
    for f in sorted(os.listdir(committed)):
        dst = os.path.join(a.dir, f)
        if not os.path.exists(dst):
            os.symlink(os.path.join(committed, f), dst)
            linked.append(f)
    env['AZG_GOLDEN_DIR'] = a.dir
    p = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_oracle_golden.py'), '-q'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = '\n'.join(p.stdout.strip().splitlines()[-12:])
    print(tail)
    if a.out:
        with open(a.out, 'w') as fh:
            fh.write('tools/check_oracle_vs_ref.py: vectors generated from the live reference (%d files): %s\n' % (len(fresh), ' '.join(fresh)))
            fh.write('committed fixtures linked in for the rest (%d files)\n\n%s\n' % (len(linked), tail))
    sys.exit(p.returncode)


if __name__ == '__main__':
    main()
