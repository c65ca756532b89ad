"""Pin the C oracle against the LIVE reference on a much larger vector set than the committed fixtures.
Build-container only (needs /root/reference):
    python tools/check_oracle_vs_ref.py [--dir /tmp/azg_big]
= tools/gen_golden.py --big DIR  (reference -> vectors)  +  pytest tests/test_oracle_golden.py with AZG_GOLDEN_DIR=DIR."""
import argparse
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dir', default='/tmp/azg_big')
    a = ap.parse_args()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    subprocess.check_call([sys.executable, os.path.join(HERE, 'gen_golden.py'), '--big', a.dir], env=env)
    env['AZG_GOLDEN_DIR'] = a.dir
    sys.exit(subprocess.call([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_oracle_golden.py'), '-q',
                              '-x'], env=env, cwd=ROOT))


if __name__ == '__main__':
    main()
