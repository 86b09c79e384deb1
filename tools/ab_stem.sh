for f in "$@"; do echo -n "$f "; ADX_LIB=$PWD/$f python3 tools/bench_stem_wgrad.py 2>&1 | grep "True" ; done
