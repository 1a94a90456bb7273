"""Generate tests/golden/config_defaults.json by RUNNING the reference's flag parser and `set_template`
(config.py:12-148,151-274) for the dataset / model codes of BASELINE.json's configs. Only the resulting flag values
leave this script. Run from the repo root:  PYTHONDONTWRITEBYTECODE=1 python tests/gen_goldens_config.py
"""
import copy
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True


def main():
    argv, cwd = sys.argv, os.getcwd()
    sys.argv = ["x"]
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        import config as ref_config   # parses argv at import (config.py:274)
    finally:
        sys.argv = argv
        os.chdir(cwd)
    out = {}
    for model_code in ("lru", "llm"):
        for ds in ("ml-100k", "beauty", "games"):
            a = copy.deepcopy(ref_config.args)
            a.model_code, a.dataset_code = model_code, ds
            ref_config.set_template(a)
            vals = {k: v for k, v in vars(a).items()
                    if isinstance(v, (int, float, str, bool, list, type(None))) and k != "device"}
            out[f"{model_code}/{ds}"] = vals
    path = os.path.join(REPO, "tests", "golden", "config_defaults.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["llm/beauty"]), "flags per case")


if __name__ == "__main__":
    main()
