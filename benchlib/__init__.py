"""Parts of bench.py (the contract, the timed region and the JSON line stay in bench.py at the repo root)."""
