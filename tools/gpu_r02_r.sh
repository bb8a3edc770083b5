#!/bin/bash
# session R: full GPU suite + smoke
cd "$(dirname "$0")/.." && export VD_QUIET=1
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
