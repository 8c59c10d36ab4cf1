"""Runs the reference's OWN parameter-file reader (oracle/_ref/ref_params: Core/src/Utils/GlobalStateParams.h + parameterFile.h
compiled from /root/reference by oracle/ref_host/Makefile) over tests/golden/ref_params/cases.py and the reference's shipped
GUI/GlobalStateParam.txt, and commits what it read: tests/golden/ref_params/expected.json {case: {member: [type, value]}}.
Members whose key is absent from the file are left out (the reference leaves them uninitialised)."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "ref_params"))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_params")


def run_reference_reader(path):
    """{member: (type, value text)} as printed by the reference's reader after readMembers()"""
    out = subprocess.run([EXE, path], capture_output=True, text=True, check=True).stdout
    ref = {}
    for line in out.split("----\n", 1)[1].split("\n"):
        if "\t" in line:
            n, t, v = line.split("\t", 2)
            ref[n] = (t, v[1:-1] if v.endswith("]") else v[1:])
    return ref


def main():
    import cases
    from hrbffusion3d_amd import config as hcfg
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_host")])
    exp = {}
    with tempfile.TemporaryDirectory() as d:
        todo = dict(cases.FILES)
        shipped = "/root/reference/GUI/GlobalStateParam.txt"
        for name, txt in todo.items():
            path = os.path.join(d, name + ".txt")
            with open(path, "w", newline="") as f:
                f.write(txt)
            ref = run_reference_reader(path)
            present = hcfg.parse_parameter_file(path)
            exp[name] = {n: list(tv) for n, tv in ref.items() if n in present}
        ref = run_reference_reader(shipped)
        exp["shipped GUI/GlobalStateParam.txt"] = {n: list(tv) for n, tv in ref.items() if n in hcfg.parse_parameter_file(shipped)}
    with open(os.path.join(HERE, "ref_params", "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    print({k: len(v) for k, v in exp.items()})


if __name__ == "__main__":
    main()
