"""Stub of `optuna` (absent offline) so that `import cogdl` works; AutoML is never exercised."""


class _Trial:  # pragma: no cover
    pass


def create_study(*a, **k):  # pragma: no cover
    raise RuntimeError("optuna stub: AutoML is not available offline")
