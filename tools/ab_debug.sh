#!/bin/bash
for f in variants/lib_*.so; do cp $f isdf_amd/libisdf_hip.so; echo "== $f"; python tools/debug_nan.py 2>&1 | grep -E "^A6|^INJ4|^ZB5|in_layer.0.weight"; done
