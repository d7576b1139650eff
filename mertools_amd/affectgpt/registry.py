"""Encoder registry with the reference's interface (MER2025/MER2025_Track23/my_affectgpt/common/registry.py:144-175,
lookups :260-290): `@registry.register_visual_encoder(name)` / `register_acoustic_encoder(name)` decorators that refuse
duplicate names, and `get_visual_encoder_class(name)` / `get_acoustic_encoder_class(name)` that return None for unknown
names.  Only the two encoder mappings are kept — builders, tasks, runners are outside the hot path."""


class Registry:
    mapping = {"visual_encoder_mapping": {}, "acoustic_encoder_mapping": {}}

    @classmethod
    def _register(cls, kind, name):
        def wrap(encoder_cls):
            table = cls.mapping[kind]
            if name in table:
                raise KeyError("Name '{}' already registered for {}.".format(name, table[name]))
            table[name] = encoder_cls
            return encoder_cls
        return wrap

    @classmethod
    def register_visual_encoder(cls, name):
        return cls._register("visual_encoder_mapping", name)

    @classmethod
    def register_acoustic_encoder(cls, name):
        return cls._register("acoustic_encoder_mapping", name)

    @classmethod
    def get_visual_encoder_class(cls, name):
        return cls.mapping["visual_encoder_mapping"].get(name, None)

    @classmethod
    def get_acoustic_encoder_class(cls, name):
        return cls.mapping["acoustic_encoder_mapping"].get(name, None)

    @classmethod
    def list_visual_encoders(cls):
        return sorted(cls.mapping["visual_encoder_mapping"].keys())

    @classmethod
    def list_acoustic_encoders(cls):
        return sorted(cls.mapping["acoustic_encoder_mapping"].keys())


registry = Registry()
