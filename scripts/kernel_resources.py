"""Registers, spills and LDS of every kernel in a device assembly file (hipcc --cuda-device-only -S): python scripts/kernel_resources.py file.s [needle]"""
import re, sys
text = open(sys.argv[1]).read()
needle = sys.argv[2] if len(sys.argv) > 2 else ""
for block in text.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block).group(1)
    if needle not in name:
        continue
    get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))
    print(f"{name[:110]:110s} vgpr {get('vgpr_count'):4d} spill {get('vgpr_spill_count'):4d} sgpr {get('sgpr_count'):4d} lds {get('group_segment_fixed_size'):6d} scratch {get('private_segment_fixed_size'):5d}")
