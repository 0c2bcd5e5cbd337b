"""Exception type used for ill-defined RadarData objects (mirrors the
reference's ``impdar.lib.ImpdarError.ImpdarError``)."""


class ImpdarError(Exception):
    pass
