_cover_ = True
fresh = 2
