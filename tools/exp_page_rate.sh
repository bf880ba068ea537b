#!/bin/bash
# page-warp driver rates with the copy workers pinned per L3 domain or left to the scheduler, process bound to the GPU's node or not
for bind in "" "--no-bind"; do for pin in 1 0 1 0; do
  echo "== bind='$bind' MICROALIGNER_COPY_PIN=$pin"
  MICROALIGNER_COPY_PIN=$pin timeout 200 python tools/page_rate.py $bind 2>&1 | head -2
done; done
