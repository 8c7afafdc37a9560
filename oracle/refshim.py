"""ORACLE (test infrastructure, build container only): import the reference's own modules from
/root/reference without running its package __init__ files (which need pytorch_lightning / DCN).
Never used on the GPU box; never imported by tests at run time (only by oracle/gen_golden.py).
SURVEY.md Appendix C."""
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "CenterNet"))


def install(dcn_cls=None):
    sys.dont_write_bytecode = True

    def stub(name, path=None):
        m = types.ModuleType(name)
        if path:
            m.__path__ = [path]
        sys.modules[name] = m
        return m

    stub("CenterNet", f"{REF}/CenterNet")
    stub("CenterNet.models", f"{REF}/CenterNet/models")
    if dcn_cls is not None:
        d = stub("DCN")
        d2 = stub("DCN.dcn_v2")
        d.dcn_v2 = d2
        d2.DCN = dcn_cls
