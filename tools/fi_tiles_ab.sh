set -e
for T in 1 2 3 4; do echo "== LPVS_FI_TILES=$T"; LPVS_FI_TILES=$T python tools/iter_time.py 20 512 8 2000 2>&1 | grep -E "group|l1 " ; done
