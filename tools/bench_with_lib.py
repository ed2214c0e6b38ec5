#!/usr/bin/env python3
"""bench.py with an EXPERIMENT build of the library: tools/bench_with_lib.py tools/libexp_X.so [bench.py flags].  A/B aid."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from simulate_2048_amd import capi  # noqa: E402

capi._default = capi.Library(sys.argv[1])
import bench  # noqa: E402

sys.argv = ['bench.py'] + sys.argv[2:]
bench.main()
