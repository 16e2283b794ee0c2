#!/usr/bin/env python3
"""Mirror the reference's classification task CONFIGS (data: prompts, generation kwargs, metric list) into
lmms_owc_amd/task_configs/.  Reads /root/reference/src/data/tasks/_classification/<dataset>/<variant>.yaml plus
assets/_default_template_yaml at generation time (the `!function` hooks are dropped: documents, prompts and targets are
handled by lmms_owc_amd/tasks.py) and writes one flat YAML per task.  `concept_semantic_similarity` is left out of the
online metric list (it needs the spaCy noun-chunk extractor; it is evaluated offline by eval_metrics.py when an
extractor is plugged in)."""
from pathlib import Path

import yaml

REF = Path("/root/reference/src/data/tasks/_classification")
OUT = Path(__file__).resolve().parent.parent / "lmms_owc_amd" / "task_configs"


class Loader(yaml.SafeLoader):
    pass


Loader.add_constructor("!function", lambda loader, node: None)


def main() -> None:
    n = 0
    for ds in sorted(p for p in REF.iterdir() if p.is_dir()):
        template = yaml.load((ds / "assets" / "_default_template_yaml").read_text(), Loader=Loader)
        for f in sorted(ds.glob("*.yaml")):
            cfg = yaml.load(f.read_text(), Loader=Loader)
            flat = {
                "task": cfg["task"],
                "dataset_path": template.get("dataset_path", f"data/{ds.name}"),
                "test_split": template.get("test_split", "test"),
                "output_type": cfg.get("output_type", template.get("output_type", "generate_until")),
                "model_specific_kwargs": cfg.get("model_specific_kwargs", {}),
                "generation_kwargs": {k: (bool(v) if k == "do_sample" else v) for k, v in cfg.get("generation_kwargs", {}).items()},
                "metric_list": [m for m in template["metric_list"] if m["metric"] != "concept_semantic_similarity"],
            }
            header = (f"# Mirror of the reference task config src/data/tasks/_classification/{ds.name}/{f.name}\n"
                      "# (+ assets/_default_template_yaml), written by tools/gen_task_configs.py; concept_semantic_similarity is\n"
                      "# evaluated offline (needs the pluggable noun-chunk extractor).\n")
            (OUT / f"{cfg['task']}.yaml").write_text(header + yaml.safe_dump(flat, sort_keys=False, allow_unicode=True, width=1000))
            n += 1
    print(f"wrote {n} task configs to {OUT}")


if __name__ == "__main__":
    main()
