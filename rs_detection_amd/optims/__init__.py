from . import optimizer, lr_scheduler  # noqa: F401  (registers SGD / AdamW / StepLR / CosineAnnealingLR)
