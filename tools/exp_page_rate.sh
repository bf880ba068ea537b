#!/bin/bash
# page-warp driver / per-page warp() rates (tools/page_rate.py), repeated
for i in 1 2 3; do
  timeout 200 python tools/page_rate.py 2>&1 | tail -4
done
