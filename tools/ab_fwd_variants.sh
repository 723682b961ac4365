#!/bin/bash
# builds the A/B partners of the pair-tile forward kernel: variants/lib_onetile.so (MODE 0 on chain.hip's one-tile kernel) and
# variants/lib_dbg.so (stage stamps for tools/timeline_fwd.py)
cd "$(dirname "$0")/.."
python tools/build_variants.py 'onetile=sed:fwd_pair.hip:s/^(bool fwd_pair_supported\(const NetLayout& l\) \{).*$/\1 (void)l; return false; }/' dbg="-DISDF_DEBUG_HOOKS=1"
