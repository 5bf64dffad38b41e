_base_ = 'plain.yaml'
epochs = 36
