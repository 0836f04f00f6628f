module github.com/shenwei356/wfa

go 1.22
