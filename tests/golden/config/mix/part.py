_base_ = ['zero.yaml', 'wipe.py']
kept = 3
