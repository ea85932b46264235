"""The six Inviwo modules as registration units and the processor factory (host/cpm_modules.*): module identifiers and
versions, the processors / ports / data formats each registers, and that every processor type the reference's
workspace instantiates can be created by class identifier with the port and property ids the workspace binds.
No GPU needed (nothing is evaluated)."""
import ctypes as C
import re
from pathlib import Path

import pytest

REF_WORKSPACE = Path("/root/reference/workspaces/CorrelatedPhotonMappingSingleVolume.inv")

MODULES = {  # ref <name>module.cpp: identifier, getVersion(), registerProcessor<...>
    "ProgressivePhotonMapping": (0, {"org.inviwo.PhotonToLightVolumeProcessorCL", "org.inviwo.ProgressivePhotonTracerCL"}),
    "LightCL": (1, {"org.inviwo.DirectionalLightSamplerCL"}),
    "RndGenMWC64X": (0, set()),
    "UniformGridCL": (1, {"org.inviwo.DynamicVolumeDifferenceAnalysis", "org.inviwo.UniformGrid3DPlayerProcessor",
                          "org.inviwo.VolumeMinMaxCLProcessor", "org.inviwo.VolumeSequencePlayer"}),
    "ImportanceSamplingCL": (1, {"org.inviwo.MinMaxUniformGrid3DImportanceCLProcessor", "org.inviwo.UniformSampleGenerator2DCL"}),
    "RadixSortCL": (0, set()),   # the module registers; its sort NODE is outside the path (the path sorts through cpm_bin / cpm_sort_*)
}
# processors of the reference's modules outside SURVEY section 8's scope: only in the -DCPM_HOST_EXTRAS build
EXTRAS = {"UniformGridCL": {"org.inviwo.UniformGrid3DExport", "org.inviwo.UniformGrid3DSequenceSelector", "org.inviwo.UniformGrid3DVectorSource"},
          "RadixSortCL": {"org.inviwo.RadixSortCL"}}
# port class identifiers the workspace spells out (.inv:465-469, 547-548, 618-619)
WORKSPACE_PORT_TYPES = {"UniformGrid3DBaseInport", "UniformGrid3DBaseOutport", "PhotonDataInport", "RecomputedPhotonIndicesInport",
                        "LightSamplesMultiInport"}


@pytest.fixture(scope="module")
def host(cpm):
    cpm.build.build_host_library()
    lib = C.CDLL(str(cpm.binding.LIB_PATH.parent / "libcpm_host.so"))
    lib.cpmh_modules_describe.restype = C.c_char_p
    lib.cpmh_factory_create.restype = C.c_char_p
    lib.cpmh_factory_create.argtypes = [C.c_char_p]
    return lib


@pytest.fixture(scope="module")
def host_extras(cpm):
    cpm.build.build_host_library(extras=True)
    lib = C.CDLL(str(cpm.binding.LIB_PATH.parent / "libcpm_host_extras.so"))
    lib.cpmh_modules_describe.restype = C.c_char_p
    lib.cpmh_factory_create.restype = C.c_char_p
    lib.cpmh_factory_create.argtypes = [C.c_char_p]
    return lib


def _parse(line):
    cid, i, o, p = line.split("|")
    return cid, set(filter(None, i[3:].split(","))), set(filter(None, o[4:].split(","))), set(filter(None, p[5:].split(",")))


def test_modules_register_the_reference_surface(host):
    seen = {}
    ports = set()
    for line in host.cpmh_modules_describe().decode().strip().splitlines():
        name, version, procs, prts, fmts = line.split("|")
        seen[name] = (int(version), set(filter(None, procs.split(","))), set(filter(None, fmts.split(","))))
        ports |= set(filter(None, prts.split(",")))
    assert set(seen) == set(MODULES)
    for name, (version, procs) in MODULES.items():
        assert seen[name][0] == version, name
        assert procs <= seen[name][1], (name, procs - seen[name][1])
    assert "u3d" in seen["UniformGridCL"][2]
    assert WORKSPACE_PORT_TYPES <= ports
    for name, procs in EXTRAS.items():                 # the product library does not register what the path does not use
        assert not (procs & seen[name][1]), (name, procs & seen[name][1])


def test_extras_build_registers_the_out_of_scope_processors(host_extras):
    seen = {}
    for line in host_extras.cpmh_modules_describe().decode().strip().splitlines():
        name, _, procs, _, _ = line.split("|")
        seen[name] = set(filter(None, procs.split(",")))
    for name, procs in EXTRAS.items():
        assert procs <= seen[name], (name, procs - seen[name])
    cid, ins, outs, _ = _parse(host_extras.cpmh_factory_create(b"org.inviwo.RadixSortCL").decode())
    assert ins == {"unsortedKeys", "unsortedData"} and outs == {"sortedData"}          # radixsortcl.cpp:194-202


def test_factory_creates_processors_by_class_identifier(host):
    line = host.cpmh_factory_create(b"org.inviwo.ProgressivePhotonTracerCL").decode()
    cid, ins, outs, props = _parse(line)
    assert cid == "org.inviwo.ProgressivePhotonTracerCL"
    assert {"volume", "recomputationImportance", "LightSamples"} <= ins and {"photons", "recomputedIndices"} <= outs
    assert {"maxIncrementalPhotonsToUpdate", "maxScatteringEvents", "radius"} <= props
    assert host.cpmh_factory_create(b"org.inviwo.RadixSortCL") == b""     # outside the path: extras build only
    assert host.cpmh_factory_create(b"org.inviwo.NoSuchProcessor") == b""


@pytest.mark.skipif(not REF_WORKSPACE.exists(), reason="reference workspace not present (GPU box)")
def test_reference_workspace_processors_are_creatable(host):
    """Every processor of the reference's own modules that CorrelatedPhotonMappingSingleVolume.inv instantiates is
    registered, and offers every port identifier the workspace connects."""
    text = REF_WORKSPACE.read_text(errors="ignore")
    ours = set().union(*(p for _, p in MODULES.values()))
    blocks = re.findall(r'<Processor type="(org\.inviwo\.[A-Za-z0-9]+)".*?</Processor>', text, flags=re.S)
    used = set(re.findall(r'<Processor type="(org\.inviwo\.[A-Za-z0-9]+)"', text))
    assert {"org.inviwo.ProgressivePhotonTracerCL", "org.inviwo.PhotonToLightVolumeProcessorCL", "org.inviwo.DirectionalLightSamplerCL",
            "org.inviwo.UniformSampleGenerator2DCL", "org.inviwo.MinMaxUniformGrid3DImportanceCLProcessor",
            "org.inviwo.VolumeMinMaxCLProcessor"} <= used
    for m in re.finditer(r'<Processor type="(org\.inviwo\.[A-Za-z0-9]+)"(.*?)</Processor>', text, flags=re.S):
        cid, body = m.group(1), m.group(2)
        if cid not in ours:
            continue  # Inviwo's own processors (VolumeSource, raycaster, canvas, ...)
        line = host.cpmh_factory_create(cid.encode()).decode()
        assert line, cid
        _, ins, outs, props = _parse(line)
        for kind, have in (("InPort", ins), ("OutPort", outs)):
            for ident in re.findall(r'<%s type="[^"]*" identifier="([^"]+)"' % kind, body):
                assert ident in have, (cid, kind, ident)
        # every property identifier the workspace serialises for this processor (composites' members included)
        for ident in re.findall(r'<Property type="[^"]*" identifier="([^"]+)"', body):
            assert ident in props, (cid, "Property", ident)
    assert blocks is not None
