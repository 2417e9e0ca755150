#!/usr/bin/env python3
"""Known answers of the reference's three LR schedules (src/open_clip_train/scheduler.py: const_lr, const_lr_cooldown, cosine_lr;
selected by --lr-scheduler, train_AT_text_only.py:384-401), produced by importing the reference's module.
Runs only in the build container.   python tests/golden/make_golden_sched.py   -> tests/golden/sched_kat.json"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


class FakeOpt:
    def __init__(self):
        self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]


def main():
    MG.install_stubs()
    from open_clip_train.scheduler import const_lr, const_lr_cooldown, cosine_lr
    steps = [0, 1, 9, 10, 11, 99, 100, 149, 150, 151, 175, 198, 199]
    cases = []
    for name, make in (("cosine", lambda o: cosine_lr(o, 1e-5, 10, 200)),
                       ("const", lambda o: const_lr(o, 1e-5, 10, 200)),
                       ("const-cooldown p=1 end=0", lambda o: const_lr_cooldown(o, 1e-5, 10, 200, 50, 1.0, 0.0)),
                       ("const-cooldown p=2 end=1e-6", lambda o: const_lr_cooldown(o, 1e-5, 10, 200, 50, 2.0, 1e-6))):
        o = FakeOpt()
        f = make(o)
        vals = []
        for s in steps:
            v = float(f(s))
            assert o.param_groups[0]["lr"] == v and o.param_groups[1]["lr"] == v
            vals.append(v)
        cases.append({"name": name, "values": vals})
    with open(os.path.join(HERE, "sched_kat.json"), "w") as fjs:
        json.dump({"base_lr": 1e-5, "warmup": 10, "total_steps": 200, "cooldown_steps": 50, "steps": steps, "cases": cases}, fjs)
    print("sched_kat.json:", [c["name"] for c in cases])


if __name__ == "__main__":
    main()
