from .clip import available_models, load, tokenize  # noqa: F401
