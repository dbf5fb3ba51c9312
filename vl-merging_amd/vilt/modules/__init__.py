from .vilt_module import ViLTransformerSS  # noqa: F401
