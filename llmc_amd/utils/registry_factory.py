"""Name -> class registries with llmc's decorator protocol (llmc/utils/registry_factory.py:1-49):
`@ALGO_REGISTRY` registers a class under its __name__, `@ALGO_REGISTRY('key')` under a given key, a second
registration of the same key raises, lookup is by item access."""


class Register(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._dict = {}

    def _add(self, key, value):
        if not callable(value):
            raise Exception(f'Error:{value} must be callable!')
        if key in self._dict:
            raise Exception(f'{key} already exists.')
        self._dict[key] = value
        return value

    def register(self, target):
        if callable(target):
            return self._add(target.__name__, target)
        return lambda x: self._add(target, x)

    __call__ = register

    def __setitem__(self, key, value):
        self._dict[key] = value

    def __getitem__(self, key):
        return self._dict[key]

    def __contains__(self, key):
        return key in self._dict

    def __str__(self):
        return str(self._dict)

    def keys(self):
        return self._dict.keys()

    def values(self):
        return self._dict.values()

    def items(self):
        return self._dict.items()


ALGO_REGISTRY = Register()
MODEL_REGISTRY = Register()
