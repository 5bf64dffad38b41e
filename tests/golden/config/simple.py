lr = 0.0025
epochs = 12
