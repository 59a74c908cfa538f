"""Test-infrastructure tooling: renders the reference's shipped model configs (examples/models/**/*.yml.j2, Jinja + YAML) and freezes
their `model_config` / `learning_config` mappings to tests/golden/reference_configs.json, so that the GPU box (which has no
/root/reference) can build models from the reference's own `class_name` + `config` (tests/test_dropin_api_gpu.py).

    python oracle/gen_config_fixture.py
"""
import json
import os

import jinja2
import yaml

REF = "/root/reference/examples/models"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_configs.json")
FILES = {
    "transducer/conformer/small": "transducer/conformer/small.yml.j2",
    "transducer/conformer/small-streaming": "transducer/conformer/small-streaming.yml.j2",
    "ctc/conformer/small": "ctc/conformer/small.yml.j2",
    "transducer/contextnet/small": "transducer/contextnet/small.yml.j2",
}


def main():
    out = {}
    for key, rel in FILES.items():
        txt = jinja2.Template(open(os.path.join(REF, rel)).read()).render(decoder_config={"vocabsize": 1000}, modeldir="/tmp/m",
                                                                          kaggle_model_handle="x", repodir="/root/reference", datadir="/tmp/d")
        doc = yaml.safe_load(txt)
        lc = doc["learning_config"]
        out[key] = {"source": "examples/models/" + rel, "model_config": doc["model_config"],
                    "learning_config": {k: lc.get(k) for k in ("optimizer_config", "gwn_config", "gradn_config", "batch_size", "ga_steps")}}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(OUT, {k: v["model_config"]["class_name"] for k, v in out.items()})


if __name__ == "__main__":
    main()
