"""Plug-in registries keyed by name, compatible with how llmc uses its own (llmc/utils/registry_factory.py):

    @ALGO_REGISTRY                 # key = class name
    class GPTQ(...): ...
    @ALGO_REGISTRY('alias')        # explicit key
    class Other(...): ...
    algo_cls = ALGO_REGISTRY[config.quant.method]

A key can be bound once; values must be callables (classes or factories).
"""
from collections.abc import Callable, Iterator


class Registry:
    """Mapping-like container with decorator registration."""

    __slots__ = ('_entries', 'kind')

    def __init__(self, kind: str = 'entry'):
        self._entries: dict = {}
        self.kind = kind

    # -- registration ---------------------------------------------------------------------------------
    def bind(self, key: str, value: Callable) -> Callable:
        if not callable(value):
            raise Exception(f'Error:{value} must be callable!')
        if key in self._entries:
            raise Exception(f'{key} already exists.')
        self._entries[key] = value
        return value

    def register(self, target):
        """Decorator: bare (`@REG`, key = __name__) or with an explicit key (`@REG('name')`)."""
        if callable(target):
            return self.bind(target.__name__, target)
        key = target

        def with_key(value):
            return self.bind(key, value)
        return with_key

    __call__ = register

    # -- lookup -----------------------------------------------------------------------------------------
    def __getitem__(self, key: str) -> Callable:
        return self._entries[key]

    def __setitem__(self, key: str, value: Callable) -> None:
        self._entries[key] = value

    def __contains__(self, key: object) -> bool:
        return key in self._entries

    def __iter__(self) -> Iterator[str]:
        return iter(self._entries)

    def __len__(self) -> int:
        return len(self._entries)

    def keys(self):
        return self._entries.keys()

    def values(self):
        return self._entries.values()

    def items(self):
        return self._entries.items()

    def get(self, key, default=None):
        return self._entries.get(key, default)

    def __repr__(self) -> str:
        return f'Registry({self.kind}: {sorted(self._entries)})'


Register = Registry  # llmc's name for the class

ALGO_REGISTRY = Registry('algorithm')
MODEL_REGISTRY = Registry('model')
