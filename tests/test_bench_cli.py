"""bench.py command line: the contract flags exist and --help renders (argparse %-formats every help string)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_help_renders_and_names_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    for flag in ('--gpus', '--steps', '--warmup', '--storage', '--mode'):
        assert flag in r.stdout, flag
