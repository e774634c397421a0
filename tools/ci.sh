#!/bin/bash
# CPU-side check (no GPU needed): build everything, run the CPU test-suite.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -x -q -m "not gpu"
echo "on an MI355X: python -m pytest tests -m gpu -x -q && python bench.py"
