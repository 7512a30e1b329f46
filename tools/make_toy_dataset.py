import sys, numpy as np, pathlib
sys.path.insert(0, "tests"); sys.path.insert(0, "audio-visual-llm_amd")
from test_data_cpu import make_set
root = pathlib.Path(sys.argv[1]); root.mkdir(parents=True, exist_ok=True)
mp, lp = make_set(root, n=4)
(root / "test.tsv").write_text(mp.read_text()); (root / "test.wrd").write_text(lp.read_text())
print("made", root)
