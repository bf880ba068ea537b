#!/bin/bash
pw() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=r['variants']['page_warp']; print(v.get('value'), v.get('pcie_gb_s_both_directions'), v.get('per_page_warp_loop',{}).get('value'), v.get('error'))"; }
echo "== lanes only, GPU_MAX_HW_QUEUES=16"; GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-pairs 0 2>/dev/null | pw
echo "== lanes only, engine log (last 1200 copies)"
AMD_LOG_LEVEL=4 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --stream-pairs 0 2>&1 | grep -o "HSA Copy copy_engine=0x[0-9a-f]*.*engineType=[0-9]*\|\"page_warp\": {[^}]*}" | sed 's/, dst=.*forceSDMA/ forceSDMA/' | tail -1500 | sort | uniq -c | sort -rn | head
