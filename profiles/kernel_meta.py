#!/usr/bin/env python3
"""Per-kernel resources from the gfx950 listings (`make -C dbde-video-cpp_amd/csrc asm`): VGPRs, SGPRs, LDS, scratch."""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "dbde-video-cpp_amd", "csrc")
for f in sys.argv[1:] or ["dbde_kernels.s", "dbde16_kernels.s"]:
    text = open(os.path.join(CSRC, f)).read()
    md = text[text.index("amdhsa.kernels:"):]
    for entry in re.split(r"\n  - \.", md)[1:]:
        get = lambda k: re.search(r"\.?%s:\s+(\S+)" % k, entry).group(1)
        print(f"{get('name')[:70]:70s} vgpr {get('vgpr_count'):>4s} sgpr {get('sgpr_count'):>4s} lds {get('group_segment_fixed_size'):>6s} "
              f"scratch {get('private_segment_fixed_size'):>4s}")
