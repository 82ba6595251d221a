"""Golden of the reference's command-line surface ("argparse flag names"): the add_argument calls of
train_final_voc.py / train_final_coco.py / tools/eval_seg_voc.py / tools/eval_seg_coco_ddp.py, read with `ast` (the scripts
cannot be imported here: LOCAL_RANK, tensorboardX, torchvision at import time) -> tests/golden/cli_flags.json.

Run:  python oracle/gen_golden_cli.py        (authoring container only; needs /root/reference)
"""
import ast
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def flags_of(path):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument" and node.args:
            name = ast.literal_eval(node.args[0])
            rec = {}
            for kw in node.keywords:
                if kw.arg == "default":
                    try:
                        rec["default"] = ast.literal_eval(kw.value)
                    except ValueError:
                        rec["default"] = ast.unparse(kw.value)
                elif kw.arg == "type":
                    rec["type"] = ast.unparse(kw.value)
                elif kw.arg == "action":
                    rec["action"] = ast.literal_eval(kw.value)
            out[name] = rec
    return out


def main():
    res = {p: flags_of(p) for p in ("train_final_voc.py", "train_final_coco.py", "tools/eval_seg_voc.py", "tools/eval_seg_coco_ddp.py")}
    for p, f in res.items():
        print(p, len(f), "flags")
    dst = os.path.join(os.path.dirname(HERE), "tests", "golden", "cli_flags.json")
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
