#!/bin/bash
# (re)build libqdg.so and the C++ test drivers from the repository root, whatever the caller's cwd
cd "$(dirname "$0")/.." || exit 1
python3 -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|Error" | head -20
ls -la --time-style=full-iso quinoa_amd/lib/libqdg.so
