#!/bin/bash
# tools/build_variant.sh <git-rev> <name>: the library of another revision as rayjoin_amd/variants/librj_<name>.so, for
# same-box A/B runs (tools/ab_bench.py; RAYJOIN_AMD_LIB selects the .so).  *.so is git-ignored and travels to the GPU box.
set -e
REV=${1:?rev}; NAME=${2:?name}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/rjvar.XXXX)
git -C $R worktree add -f $T $REV -q
make -C $T/rayjoin_amd/csrc -j4 > $T/build.log 2>&1 || { tail $T/build.log; exit 1; }
mkdir -p $R/rayjoin_amd/variants && cp $T/rayjoin_amd/librayjoin_amd.so $R/rayjoin_amd/variants/librj_$NAME.so
git -C $R worktree remove --force $T
ls -la $R/rayjoin_amd/variants/librj_$NAME.so
