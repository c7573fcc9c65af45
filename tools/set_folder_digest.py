"""Write the digest a bench line carries (`outputs_check.digest`) into bench.FOLDER_DIGEST_EXPECTED.
python tools/set_folder_digest.py <bench line .json>   -- the GPU test test_folder_512_clips_one_dense_pass then holds its
verified outputs to that value; a mismatch there means the bench's outputs are not the verified ones."""
import json, os, re, sys
line = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
d = int(line["outputs_check"]["digest"])
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
s = open(p).read()
s2 = re.sub(r"^FOLDER_DIGEST_EXPECTED = .*$", f"FOLDER_DIGEST_EXPECTED = {d}", s, count=1, flags=re.M)
assert s2 != s or f"= {d}" in s
open(p, "w").write(s2)
print("FOLDER_DIGEST_EXPECTED =", d)
