#!/bin/bash
for c in 8192 16384 32768 65536 131072; do echo "chunk $c:"; HG_CHUNK_ROWS=$c python tools/bench_configs.py 2>&1 | grep '"config": 4' | cut -c1-200; done
for d in 0 2; do echo "HG_DUO=$d:"; HG_DUO=$d python tools/bench_configs.py 2>&1 | grep '"config"' | cut -c1-200; done
