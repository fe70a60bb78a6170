#!/bin/bash
# A/B: columns per wavefront of the block-Thomas sweep (MIMSEM_THOMAS_CPW = 4, 2, 1) on the bench column extras
for c in 4 2 1 0; do echo "== MIMSEM_THOMAS_CPW=$c"; MIMSEM_THOMAS_CPW=$c python bench.py --no-cpu --no-sw --cold 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['column']; print({k:round(v,3) for k,v in d.items() if 'ms' in k and not isinstance(v,dict)})"; done
