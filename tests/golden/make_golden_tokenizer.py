"""Golden token ids made by RUNNING THE REFERENCE's SimpleTokenizer here (src/open_clip/tokenizer.py, loaded by file
path with an identity stub for ftfy, which is absent and only repairs mojibake).  Stores inputs and OUTPUT ids only in
tests/golden/tokenizer.json.  Run from the repo root: python tests/golden/make_golden_tokenizer.py"""
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [
    "a photo of a dog but not a cat", "A photo of a Dog, but NOT a cat.", "  hello   world!! ",
    "it's the cat's toy; don't, we'll, I'm, they've, you're, he'd", "naive cafe 12345 3.14 #hashtag @user",
    "naïve café 中文 \U0001F600", "x" * 300, "<start_of_text> weird <end_of_text>", "&amp;amp; &lt;tag&gt;", "",
    " ".join(["word"] * 100), "There is no umbrella; a ZEBRA (large) sits here!", "tabs\tand\nnewlines",
    "a man riding a wave on top of a surfboard but no people are watching",
]
ftfy = types.ModuleType("ftfy")
ftfy.fix_text = lambda x: x
sys.modules["ftfy"] = ftfy
spec = importlib.util.spec_from_file_location("ref_tok", "/root/reference/src/open_clip/tokenizer.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
tok = ref.SimpleTokenizer()
out = dict(vocab_size=tok.vocab_size, sot=tok.sot_token_id, eot=tok.eot_token_id,
           cases=[dict(text=c, ids=tok.encode(c), ctx77=tok(c)[0].tolist(), ctx16=tok(c, 16)[0].tolist(),
                       decoded=tok.decode(tok.encode(c))) for c in CASES])
json.dump(out, open(os.path.join(HERE, "tokenizer.json"), "w"), ensure_ascii=True)
print("wrote tokenizer.json:", len(CASES), "cases")
