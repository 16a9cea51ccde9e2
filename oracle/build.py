"""Builds the oracle's compiled parts (run by __graft_entry__.build()): the C++/OpenMP CPU baseline library oracle/cpu.
Test infrastructure only - the product package never loads it."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cpu import build  # noqa: E402

if __name__ == '__main__':
    print('built', build(force='--force' in sys.argv))
