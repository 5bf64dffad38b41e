_base_ = ['part.py', 'deep/extra.yaml', '../plain.yaml']
omega = 9
