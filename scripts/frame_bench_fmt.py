import json
import sys
print("  ".join("%s: %.3f ms %.3e/s" % (r["frame"], r["ms_per_launch"], r["frame_solves_per_s"]) for r in map(json.loads, sys.stdin)))
