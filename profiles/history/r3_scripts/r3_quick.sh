#!/bin/bash
# usage: r3_quick.sh "<pytest -k expression>" [file]
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; echo "build rc=$?"
timeout -k 5 900 python -m pytest ${2:-tests} -m gpu -x -q -k "$1" 2>&1 | tail -15
