#!/bin/bash
# session R: full GPU suite + smoke
cd "$(dirname "$0")/.." && export VD_QUIET=1
timeout 1700 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
