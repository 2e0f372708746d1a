cd practical-collab-perception_amd
echo "== NW auto (8 where cout_pad % 128 == 0)"; timeout 300 python tools/bench_w4c.py 20 2>&1 | tail -14
echo "== NW forced 4"; PCP_WINO4C_NW=4 timeout 300 python tools/bench_w4c.py 20 2>&1 | tail -14 | head -10
