"""Per-kernel resource table (VGPRs, SGPR/VGPR spills, scratch, LDS) from the metadata of an `hipcc -S` listing.
usage: isa_meta.py FILE.s [name-filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    if flt not in name:
        continue
    print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} sspill {g('sgpr_spill_count'):>4s} vspill {g('vgpr_spill_count'):>4s} "
          f"scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
