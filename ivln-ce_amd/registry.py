"""`baseline_registry` with the API the reference uses:
register_policy / register_trainer(name=) / register_env(name=) / register_obs_transformer(),
get_policy / get_trainer / get_env / get_obs_transformer.

When habitat_baselines is importable its own registry is used, so the plugins below drop into an
existing Habitat installation unchanged (ivlnce_baselines/common/obs_transforms.py:30,
models/map_cma_policy.py:28, trainers/dagger_trainer.py:237 register through the same object).
"""
from typing import Any, Callable, Dict, Optional

try:  # pragma: no cover - habitat is not in the build image
    from habitat_baselines.common.baseline_registry import baseline_registry  # type: ignore

    HAVE_HABITAT = True
except Exception:  # noqa: BLE001
    HAVE_HABITAT = False

    class _Registry:
        def __init__(self):
            self._m: Dict[str, Dict[str, Any]] = {
                "policy": {}, "trainer": {}, "env": {}, "obs_transformer": {},
            }

        def _register(self, kind: str, to_register: Optional[Callable], name: Optional[str]):
            def wrap(cls):
                self._m[kind][name or cls.__name__] = cls
                return cls

            return wrap if to_register is None else wrap(to_register)

        def register_policy(self, to_register=None, *, name: Optional[str] = None):
            return self._register("policy", to_register, name)

        def register_trainer(self, to_register=None, *, name: Optional[str] = None):
            return self._register("trainer", to_register, name)

        def register_env(self, to_register=None, *, name: Optional[str] = None):
            return self._register("env", to_register, name)

        def register_obs_transformer(self, to_register=None, *, name: Optional[str] = None):
            return self._register("obs_transformer", to_register, name)

        def get_policy(self, name):
            return self._m["policy"].get(name)

        def get_trainer(self, name):
            return self._m["trainer"].get(name)

        def get_env(self, name):
            return self._m["env"].get(name)

        def get_obs_transformer(self, name):
            return self._m["obs_transformer"].get(name)

    baseline_registry = _Registry()
