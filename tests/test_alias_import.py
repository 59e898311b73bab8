def test_reference_import_path_resolves_to_hip_package():
    from dia.model import Dia, ComputeDtype
    from dia.config import DiaConfig
    import dia_hip.model as m
    assert Dia is m.Dia and ComputeDtype.BFLOAT16.value == "bfloat16"
    assert DiaConfig.__module__ == "dia_hip.config"
