"""Compiles the oracle's C restatement (gcc only, no GPU code).  ORACLE = test infrastructure."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
SRC = os.path.join(HERE, "mvndst_oracle.c")
LIB = os.path.join(OUT, "libmvndst_oracle.so")


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
