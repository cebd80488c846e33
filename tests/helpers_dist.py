import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def tables_for(first, n):
    from helpers import counter_tables
    return counter_tables(n, 2024, episode=1, env_id_base=first)[0]
