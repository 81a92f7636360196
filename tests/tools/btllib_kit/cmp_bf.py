"""compares two btllib .bf files: header fields (key order and version suffix aside) and the raw array"""
import sys


def load(path):
    raw = open(path, "rb").read()
    end = raw.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    hdr = {}
    lines = raw[:end].decode().splitlines()
    hdr["signature"] = lines[0].split("_v")[0]
    for l in lines[1:]:
        if "=" in l:
            k, v = [x.strip() for x in l.split("=", 1)]
            hdr[k] = v.strip('"')
    return hdr, raw[end:]


a, b = load(sys.argv[1]), load(sys.argv[2])
ok = True
for key in sorted(set(a[0]) | set(b[0])):
    if a[0].get(key) != b[0].get(key):
        print("filter  header field %s: %r vs %r" % (key, a[0].get(key), b[0].get(key)))
        ok = False
if a[1] != b[1]:
    n = sum(1 for x, y in zip(a[1], b[1]) if x != y)
    print("filter  arrays DIFFER: %d of %d bytes (sizes %d / %d)" % (n, len(a[1]), len(a[1]), len(b[1])))
    ok = False
else:
    print("filter  %s vs %s: header fields and %d array bytes identical" % (sys.argv[1].split("/")[-1], sys.argv[2].split("/")[-1], len(a[1])))
sys.exit(0 if ok else 1)
