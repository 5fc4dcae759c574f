#!/bin/bash
# Run the micro-benchmarks (prebuilt by `make -C tools/micro`) and print their reports.
D=$(cd "$(dirname "$0")" && pwd)
for b in mfma_peak mfma_store launch_rate coissue tchain mfma_round mall_pc interleave dot2 f16_ovfl; do
  echo "=== $b"
  timeout 120 "$D/bin/$b"
done
