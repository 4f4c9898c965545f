"""Print registers / scratch per kernel from a --save-temps gfx950 .s file."""
import re, sys
s = open(sys.argv[1]).read()
for blk in s.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    print("%-52s vgpr %4s agpr %4s spill %4s scratch %5s lds %6s sgpr %3s" % (
        g("name").group(1)[:52], g("vgpr_count").group(1), blk.split()[0], g("vgpr_spill_count").group(1),
        g("private_segment_fixed_size").group(1), g("group_segment_fixed_size").group(1), g("sgpr_count").group(1)))
